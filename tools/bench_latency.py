"""Latency of the headline step (canonicalize + invert, 224x224x3, C8) at small batch sizes, eager and as a replayed hipGraph:
python tools/bench_latency.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    can = bench.build_canonicalizer(dev)
    for B in [int(v) for v in os.environ.get("EQA_LAT_BATCHES", "1,4,16,64,256").split(",")]:
        x = torch.randn(B, 3, 224, 224, device=dev)
        f = torch.randn(B, 3, 224, 224, device=dev)

        def step():
            y = can(x)
            return y, can.invert_canonicalization(f, induced_rep_type="scalar")

        with torch.no_grad():
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            reps = 50 if B <= 64 else 10
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / reps
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(3):
                    step()
                s.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    out = step()
            torch.cuda.synchronize()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / reps
        print(f"B={B:4d}: eager {eager*1e3:7.3f} ms ({B/eager:9.0f} img/s)   hipGraph replay {graph*1e3:7.3f} ms ({B/graph:9.0f} img/s)")


if __name__ == "__main__":
    main()
