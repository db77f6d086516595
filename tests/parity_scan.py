"""Distance of the canonicalizing transform / the inverse action to the oracle (torch CPU affine_grid + grid_sample under the
reference's op sequence), per group element and frame size, on unit-variance white noise -- the numbers behind the tolerances of
tests/test_gpu_parity.py (test infrastructure: it lives under tests/ because it calls the oracle).  On the GPU box, repo root:   python tests/parity_scan.py
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiadapt_amd import ops                                      # noqa: E402
from equiadapt_amd.images.utils import device_tables               # noqa: E402
from oracle import image_ops as io                                 # noqa: E402

dev = torch.device("cuda:0")


def err(got, want):
    d = (got.detach().cpu().double() - want.double()).abs()
    return d.max().item(), d.pow(2).mean().sqrt().item()


def main():
    torch.manual_seed(0)
    for (C, H, W), N, refl in [((3, 224, 224), 8, False), ((3, 224, 224), 4, True), ((3, 64, 64), 8, True), ((3, 33, 47), 8, False),
                               ((3, 1024, 1024), 4, True), ((3, 512, 384), 8, False)]:
        G = 2 * N if refl else N
        x = torch.randn(G, C, H, W)
        gidx = torch.arange(G)
        ang = io.group_angles(N)
        rot = (torch.cat([ang, ang]) if refl else ang)[gidx]
        ref = (gidx >= N).float() if refl else None
        pad = math.ceil(W * 0.5)
        want = io.canonicalize_images(x, rot, ref, (C, H, W))
        theta, flags = device_tables("canonicalize", N, refl, (H + 2 * pad, W + 2 * pad), dev)
        got = ops.canon_transform(x.to(dev), gidx.to(dev, torch.int32), theta, flags, pad)
        per = [err(got[g:g + 1], want[g:g + 1]) for g in range(G)]
        print(f"canonicalize {C}x{H}x{W} N={N} refl={refl}: max over elements {max(p[0] for p in per):.3e}  rms {max(p[1] for p in per):.3e}   per element max: "
              + " ".join(f"{p[0]:.1e}" for p in per), flush=True)
        # the inverse action on a scalar feature map of the same size
        want = io.invert_action(x, rot, ref, N, G, "scalar")
        th, fl = device_tables("invert", N, refl, (H, W), dev)[:2]
        got = ops.invert_action(x.to(dev), gidx.to(dev, torch.int32), th, fl, None)
        per = [err(got[g:g + 1], want[g:g + 1]) for g in range(G)]
        print(f"invert       {C}x{H}x{W} N={N} refl={refl}: max over elements {max(p[0] for p in per):.3e}  rms {max(p[1] for p in per):.3e}   per element max: "
              + " ".join(f"{p[0]:.1e}" for p in per), flush=True)


if __name__ == "__main__":
    main()
