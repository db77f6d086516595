"""eqa_mask_action_nearest alone: uint8 masks of 1024 x 1024, D4 elements, 3 masks per image (config 5): python tools/kbench_mask.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from equiadapt_amd import ops
from equiadapt_amd.images import geometry

dev = torch.device("cuda:0")
rth = geometry.mask_rotation_table((-geometry.group_angles(4)).tolist(), (1024, 1024)).to(dev)
fl = torch.full((4,), geometry.FLIP_SRC, dtype=torch.int32, device=dev)
for B in (1, 2, 4, 8, 32):
    m = (torch.rand(3 * B, 1024, 1024, device=dev) > 0.5).to(torch.uint8)
    e = torch.randint(0, 4, (3 * B,), device=dev, dtype=torch.int32)
    for _ in range(5):
        ops.mask_action_nearest(m, e, rth, fl)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    a.record()
    for _ in range(n):
        ops.mask_action_nearest(m, e, rth, fl)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    print(f"B={B:3d} ({3 * B:3d} masks): {us:7.1f} us  {2 * m.numel() / us / 1e6:6.2f} TB/s  frac {2 * m.numel() / us / 1e6 / 8:.3f}")
