"""eqa_crop_resize_aa alone (HIP events): 224 -> crop 180 -> 96 at B = 256 (input fits the Infinity Cache when looped) and B = 1024
(it does not); config 5's 1024 -> 128 (17 taps) at B = 4 / 32 over a ring of inputs, with the result checked against torch's
antialiased interpolate.  EQA_AA_STAGED=0 selects the gather form, EQA_AA_STREAM=0 the row-staged wide-filter kernel.
python tools/kbench_aa.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from equiadapt_amd import ops
from equiadapt_amd.images import geometry
from equiadapt_amd.images.transforms import resized_output_size

dev = torch.device("cuda")
for B in (256, 1024):
    H = W = 224
    crop = (math.ceil(H * 0.8), math.ceil(W * 0.8))
    out_hw = tuple(resized_output_size(crop, 96))
    tabs = tuple(v.to(dev) if isinstance(v, torch.Tensor) else v for v in geometry.aa_resize_tables((H, W), crop, out_hw))
    x = torch.randn(B, 3, H, W, device=dev)
    for _ in range(5):
        ops.crop_resize_aa(x, tabs, out_hw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.crop_resize_aa(x, tabs, out_hw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    alg = B * 3 * (crop[0] * crop[1] + out_hw[0] * out_hw[1]) * 4
    print(f"B={B} staged={os.environ.get('EQA_AA_STAGED', '1')} {us:.1f} us  {alg / us / 1e6:.2f} TB/s algorithmic")

import torch.nn.functional as F
for B in (4, 32):
    H = W = 1024
    out_hw = (128, 128)
    tabs = tuple(v.to(dev) if isinstance(v, torch.Tensor) else v for v in geometry.aa_resize_tables((H, W), (H, W), out_hw))
    xs = [torch.randn(B, 3, H, W, device=dev) for _ in range(max(2, 700 // (B * 12)))]     # ring: no launch finds its input cached
    got = ops.crop_resize_aa(xs[0], tabs, out_hw)
    want = F.interpolate(xs[0].double(), size=out_hw, mode="bilinear", antialias=True, align_corners=False)
    err = (got.double() - want).abs().max().item()
    for i in range(5):
        ops.crop_resize_aa(xs[i % len(xs)], tabs, out_hw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for i in range(n):
        ops.crop_resize_aa(xs[i % len(xs)], tabs, out_hw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    alg = B * 3 * (H * W + out_hw[0] * out_hw[1]) * 4
    print(f"1024 -> 128, B={B} stream={os.environ.get('EQA_AA_STREAM', '1')} {us:.1f} us  {alg / us / 1e6:.2f} TB/s algorithmic   max |err| vs fp64 interpolate {err:.2e}")
