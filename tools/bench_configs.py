"""Throughput of the hot path on every BASELINE.json config shape (1 GPU), with the CPU oracle timed beside it on a bounded
sample:  python tools/bench_configs.py [--out profiles/r01/configs.json]

  cfg1  CIFAR-10 shape 32x32x3, C4, GroupEquivariantImageCanonicalization + CustomEquivariantNetwork: canonicalize + invert
  cfg2  224x224x3, C8, ESCNN-shaped network (the headline: bench.py is authoritative, repeated here for the table)
  cfg4  ModelNet40 shape, 1024 points, SO(3): VNSmall -> Gram-Schmidt -> rotate (the reference defines no invert for clouds)
  cfg5  COCO shape 1024x1024x3, D4, OptimizedGroupEquivariantImageCanonicalization + ConvNetwork(k7,16ch,3 layers,128):
        canonicalize with mask targets + invert_canonicalization of a mask-shaped scalar output
(cfg3 = cfg2 on 8 GPUs: bench.py --gpus 8 under torch.distributed.run.)
"""
import argparse
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import equiadapt_amd as ea  # noqa: E402
import bench  # noqa: E402  (the CPU-oracle legs live in bench.py next to its cpu_baseline)


def gpu_time(fn, reps=20, warm=5, rounds=3):
    """Best of `rounds` timing loops (a one-time library initialisation inside one loop must not count as steady state)."""
    best = float("inf")
    with torch.no_grad():
        for _ in range(warm):
            fn()
        for _ in range(rounds):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {}

    # ---- cfg1: CIFAR-10 shape, C4
    torch.manual_seed(2)
    net = ea.CustomEquivariantNetwork((3, 32, 32), 8, 5, "rotation", 4, 2, device="cpu")
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=1.0, resize_shape=32)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 32, 32))
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    can = can.to(dev).eval()
    for B in (128, 8192):
        x = torch.randn(B, 3, 32, 32, device=dev)
        f = torch.randn(B, 3, 32, 32, device=dev)
        dt = gpu_time(lambda: (can(x), can.invert_canonicalization(f, induced_rep_type="scalar")))
        res[f"cfg1_cifar32_c4_B{B}"] = {"images_s": B / dt, "ms": dt * 1e3}
    res["cfg1_cifar32_c4_cpu_oracle"] = bench.cpu_baseline_config("cfg1", {"sd": sd})

    # ---- cfg2: headline (same construction as bench.py)
    can2 = bench.build_canonicalizer(dev)
    x = torch.randn(256, 3, 224, 224, device=dev)
    f = torch.randn(256, 3, 224, 224, device=dev)
    dt = gpu_time(lambda: (can2(x), can2.invert_canonicalization(f, induced_rep_type="scalar")), reps=10)
    res["cfg2_224_c8_B256"] = {"images_s": 256 / dt, "ms": dt * 1e3}
    del can2, x, f

    # ---- cfg4: ModelNet40 shape
    hp4 = types.SimpleNamespace(n_knn=20, pooling="mean")
    torch.manual_seed(2)
    vn = ea.VNSmall(hp4)
    sd4 = {k: v.clone() for k, v in vn.state_dict().items()}
    can4 = ea.EquivariantPointcloudCanonicalization(vn, hp4).to(dev).eval()
    for B in (64, 2048):
        pc = torch.randn(B, 3, 1024, device=dev)
        dt = gpu_time(lambda: can4(pc))
        res[f"cfg4_modelnet1024_so3_B{B}"] = {"clouds_s": B / dt, "ms": dt * 1e3}
    res["cfg4_modelnet1024_so3_cpu_oracle"] = bench.cpu_baseline_config("cfg4", {"sd": sd4})

    # ---- cfg5: COCO shape, D4, optimised canonicalizer, masks
    torch.manual_seed(2)
    net5 = ea.ConvNetwork((3, 128, 128), out_channels=16, kernel_size=7, num_layers=3, out_vector_size=128)
    hp5 = types.SimpleNamespace(beta=1.0, input_crop_ratio=1.0, resize_shape=128, group_type="roto-reflection", num_rotations=4,
                                artifact_err_wt=0.0, learn_ref_vec=False)
    can5 = ea.OptimizedGroupEquivariantImageCanonicalization(net5, hp5, (3, 1024, 1024)).to(dev).eval()
    for B in (4, 32):
        x = torch.randn(B, 3, 1024, 1024, device=dev)
        pred = torch.randn(B, 1, 1024, 1024, device=dev)
        masks = [(torch.rand(3, 1024, 1024, device=dev) > 0.5).to(torch.uint8) for _ in range(B)]
        boxes = [torch.tensor([[10.0, 20.0, 200.0, 300.0]] * 3, device=dev) for _ in range(B)]

        def step5():
            targets = [{"boxes": b.clone(), "masks": m} for b, m in zip(boxes, masks)]
            y, t = can5(x, targets)
            return y, t, can5.invert_canonicalization(pred, induced_rep_type="scalar")
        dt = gpu_time(step5, reps=10)
        res[f"cfg5_coco1024_d4_B{B}"] = {"images_s": B / dt, "ms": dt * 1e3, "note": "3 uint8 masks + 3 boxes per image as targets"}
    res["cfg5_coco1024_d4_cpu_oracle_transform_only"] = bench.cpu_baseline_config("cfg5", {})

    for k, v in res.items():
        print(k, json.dumps(v))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
