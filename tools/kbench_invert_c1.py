"""configs[4]'s inverse on a one-channel output: invert_action on (B, 1, 1024, 1024), D4, over a cache-cold ring, against
torch's copy of the same bytes.  python tools/kbench_invert_c1.py [--batch 32] [--reps 40]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiadapt_amd import _lib, ops  # noqa: E402
from equiadapt_amd.images.utils import device_tables  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--hw", type=int, default=1024)
    ap.add_argument("--channels", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, S, C = args.batch, args.hw, args.channels
    th, fl, _ = device_tables("invert", 4, True, (S, S), dev)
    nbytes = 2 * B * C * S * S * 4
    ring = max(2, int(1.3e9 // nbytes))
    xs = [torch.randn(B, C, S, S, device=dev) for _ in range(ring)]
    ys = [torch.empty_like(xs[0]) for _ in range(ring)]
    it = [0]

    def cold(fn):
        def run():
            it[0] = (it[0] + 1) % ring
            fn(xs[it[0]], ys[it[0]])
        return run
    ms = timeit(cold(lambda a, b: b.copy_(a)), args.reps)
    print(f"B = {B}, C = {C}, {S} x {S}, ring of {ring}: torch copy_  {ms*1e3:7.1f} us  {nbytes/ms/1e6:7.1f} GB/s")
    gen = torch.Generator().manual_seed(3)
    lib = _lib.load()
    keep = lib.eqa_get_option(3)
    for name, g in [("identity", torch.zeros(B)), ("quarter turn", torch.ones(B)), ("half turn", torch.full((B,), 2)),
                    ("flip", torch.full((B,), 4)), ("flip + quarter turn", torch.full((B,), 5)),
                    ("random D4", torch.randint(0, 8, (B,), generator=gen))]:
        gidx = g.to(dev, torch.int32)
        row = []
        for tiles in (0, 2, 4):       # eqa_set_option(3, .): tiles per block of the one-channel kernel (0 = the general kernel)
            lib.eqa_set_option(3, tiles)
            ms = timeit(cold(lambda a, b: ops.invert_action(a, gidx, th, fl, None)), args.reps)
            row.append(f"{tiles} tiles/block {ms*1e3:6.1f} us {nbytes/ms/1e6:7.1f} GB/s")
        lib.eqa_set_option(3, keep)
        print(f"  invert_action {name:20s} " + " | ".join(row))


if __name__ == "__main__":
    main()
