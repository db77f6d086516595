"""Training-mode step of the headline canonicalizer alone (no prediction network): forward in train(), a loss that reaches the
network through the canonical image (d/d rotation) and through the prior, backward.  python tools/bench_train.py [--batch 64]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    can = bench.build_canonicalizer(dev).train()
    opt = torch.optim.SGD(can.parameters(), lr=1e-3)
    x = torch.randn(args.batch, 3, 224, 224, device=dev)
    w = torch.randn(args.batch, 3, 224, 224, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        y = can(x)
        loss = (y * w).mean() + can.get_prior_regularization_loss()
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"canonicalizer training step B={args.batch}: {dt*1e3:.1f} ms  {args.batch/dt:.0f} img/s  "
          f"peak mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB")


if __name__ == "__main__":
    main()
