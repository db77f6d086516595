"""Data-parallel training of a canonicalizer + prediction network on synthetic CIFAR-shaped data.

    python examples/train_dp.py --steps 20                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_dp.py

One process per GPU; gradients of canonicalizer + prediction network are all-reduced by DDP over RCCL (backend "nccl"
on ROCm).  Step semantics follow the reference's Lightning module (examples/images/classification/model.py:59-127,
184-239) through ``equiadapt_amd.training``.  The prediction network is a plain PyTorch ResNet-style CNN standing in for
the unmodified third-party model; data are random images with labels derived from the image content, so the loss moves.
"""
import argparse
import os
import sys
import time
import types

import torch
import torch.distributed as dist
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import equiadapt_amd as ea  # noqa: E402
from equiadapt_amd import training as tr  # noqa: E402


def small_resnet(num_classes: int = 10) -> nn.Module:
    def block(cin, cout, stride):
        return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                             nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))

    return nn.Sequential(block(3, 32, 1), block(32, 64, 2), block(64, 128, 2), nn.AdaptiveAvgPool2d(1), nn.Flatten(),
                         nn.Linear(128, num_classes))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128, help="per GPU")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--group", default="rotation", choices=["rotation", "roto-reflection"])
    ap.add_argument("--num-rotations", type=int, default=4)
    ap.add_argument("--prior-weight", type=float, default=100.0)
    ap.add_argument("--force-ddp", action="store_true", help="wrap in DistributedDataParallel even with one process")
    ap.add_argument("--net", default="custom", choices=["custom", "escnn"],
                    help="canonicalization network: CustomEquivariantNetwork, or the ESCNN-shaped network (Winograd training path)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" in os.environ:  # under torch.distributed.run, also with a single process (DDP is then still exercised)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(0)

    S = args.size
    if args.net == "escnn":
        net = ea.ESCNNEquivariantNetwork((3, S // 2, S // 2), out_channels=32, kernel_size=5, group_type=args.group,
                                         num_rotations=args.num_rotations, num_layers=3)
    else:
        net = ea.CustomEquivariantNetwork((3, S // 2, S // 2), 8, 5, args.group, args.num_rotations, 2, device="cpu")
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=0.8, resize_shape=S // 2)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, S, S))
    model = tr.CanonicalizedClassifier(can, small_resnet(), tr.LossWeights(1.0, args.prior_weight)).to(dev)
    ddp = tr.wrap_ddp(model, dev, force=args.force_ddp)
    opt, _ = tr.configure_optimizer(model, 1e-3, 1e-3, kind="adamw")

    gen = torch.Generator().manual_seed(100 + rank)
    t0 = time.perf_counter()
    for step in range(args.steps):
        x = torch.randn(args.batch, 3, S, S, generator=gen).to(dev)
        y = (x.mean(dim=(1, 2, 3)) * 40).long().clamp(-5, 4) + 5          # labels that depend on the image
        out = tr.train_step(ddp, opt, x, y)
        if step % 5 == 0 or step == args.steps - 1:      # metrics only where they are printed: reduce_metrics() ends in a
            m = tr.reduce_metrics({k: v for k, v in out.items() if v.dim() == 0})   # .tolist() = one host sync
        if rank == 0 and (step % 5 == 0 or step == args.steps - 1):
            print(f"step {step:3d}  " + "  ".join(f"{k}={v:.4f}" for k, v in m.items()), flush=True)
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.perf_counter() - t0
        print(f"{args.steps} steps, {args.batch * world * args.steps / dt:.0f} img/s over {world} GPU(s)")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
