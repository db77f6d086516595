"""rocprofv3 drivers for the non-headline inference configurations (a few steady-state steps each): python tools/prof_cfg.py NAME
  cfg1       CIFAR-10 shape 32x32x3, C4, CustomEquivariantNetwork(8 fields, k5, 2 layers), B = 8192: canonicalize + invert
  c4_64      64x64x3, C4, ESCNNEquivariantNetwork(32 fields, k5, 3 layers) on a 64x64 crop (Winograd path: 48x48 FFT tiles do not fit), B = 256
  cloud_k16  ModelNet40 shape with k = 16 neighbours (VNSmall, mean pooling), B = 64"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import equiadapt_amd as ea  # noqa: E402

name = sys.argv[1]
dev = torch.device("cuda:0")
torch.manual_seed(2)
reps = int(os.environ.get("REPS", "10"))
with torch.no_grad():
    if name == "cfg1":
        net = ea.CustomEquivariantNetwork((3, 32, 32), 8, 5, "rotation", 4, 2, device="cpu")
        hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=1.0, resize_shape=32)
        can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 32, 32)).to(dev).eval()
        x, f = torch.randn(8192, 3, 32, 32, device=dev), torch.randn(8192, 3, 32, 32, device=dev)
        for _ in range(reps):
            can(x)
            can.invert_canonicalization(f, induced_rep_type="scalar")
    elif name == "c4_64":
        net = ea.ESCNNEquivariantNetwork((3, 64, 64), out_channels=32, kernel_size=5, group_type="rotation", num_rotations=4, num_layers=3)
        hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=1.0, resize_shape=64)
        can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 64, 64)).to(dev).eval()
        x, f = torch.randn(256, 3, 64, 64, device=dev), torch.randn(256, 3, 64, 64, device=dev)
        for _ in range(reps):
            can(x)
            can.invert_canonicalization(f, induced_rep_type="scalar")
    elif name == "cloud_k16":
        hp = types.SimpleNamespace(n_knn=16, pooling="mean")
        can = ea.EquivariantPointcloudCanonicalization(ea.VNSmall(hp), hp).to(dev).eval()
        pc = torch.randn(64, 3, 1024, device=dev)
        for _ in range(reps):
            can(pc)
    else:
        raise SystemExit(f"unknown configuration {name}")
torch.cuda.synchronize()
