"""Throughput of the hot path on the BASELINE.json config shapes other than the headline (cfg1 CIFAR-10 32x32 C4, cfg4 ModelNet40
SO(3), cfg5 COCO-shape D4 with masks), with the CPU oracle timed beside each on a bounded sample:

    python tools/bench_configs.py [--out profiles/r02/configs.json]

The legs live in bench.py (`leg_configs`), whose default run carries them in its JSON line under "configs"; this tool only
runs them alone on one GPU.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    comm = bench.Comm(dry_run=False)
    res = bench.leg_configs(comm, with_cpu=True)
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
