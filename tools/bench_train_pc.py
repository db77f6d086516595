"""Training-mode step of the point-cloud canonicalizer alone (VNSmall -> Gram-Schmidt -> rotate), forward + backward:
python tools/bench_train_pc.py [--batch 64]"""
import argparse
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import equiadapt_amd as ea  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hp = types.SimpleNamespace(n_knn=20, pooling="mean")
    torch.manual_seed(2)
    can = ea.EquivariantPointcloudCanonicalization(ea.VNSmall(hp), hp).to(dev).train()
    opt = torch.optim.SGD(can.parameters(), lr=1e-3)
    x = torch.randn(args.batch, 3, 1024, device=dev)
    w = torch.randn(args.batch, 3, 1024, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        y = can(x)
        loss = (y * w).mean() + can.get_prior_regularization_loss()
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"point-cloud canonicalizer training step B={args.batch}: {dt*1e3:.2f} ms  {args.batch/dt:.0f} clouds/s  "
          f"peak mem {torch.cuda.max_memory_allocated()/1e9:.2f} GB")


if __name__ == "__main__":
    main()
