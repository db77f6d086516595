"""Which spelling of the fp32 arithmetic does THIS host's torch CPU build use in F.affine_grid / F.grid_sample / torch.linspace?
(Test infrastructure: the HIP kernel pins one spelling -- csrc/group_action.hip lin_m1_p1 / sample_point / blend4 -- and the oracle is
torch on the CPU of whatever box runs the tests.)  Counts the values that differ from each candidate; fma is emulated through
float64 (exact product, one extra rounding: a handful of double-rounding mismatches are possible, thousands mean a different form).

    python tests/cpu_arith_probe.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import image_ops as io  # noqa: E402

f32 = np.float32


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def main():
    print(torch.__version__, torch.backends.cpu.get_cpu_capability(), "threads", torch.get_num_threads())
    for steps in (448, 94, 2048):
        step = f32(2.0) / f32(steps - 1)
        idx = np.arange(steps)
        lo = idx < (steps >> 1)
        k = np.where(lo, idx, steps - 1 - idx).astype(f32)
        mine = np.where(lo, fma(step, k, f32(-1)), fma(-step, k, f32(1))).astype(f32)
        print(f"linspace({steps}): fma form differs in {int((mine != torch.linspace(-1, 1, steps).numpy()).sum())}")
    torch.manual_seed(0)
    for H, W, deg in ((448, 448, 45.0), (448, 448, 135.0), (96, 128, 30.0)):
        img = torch.randn(1, 1, H, W)
        center = torch.tensor([[(W - 1) / 2, (H - 1) / 2]])
        theta = io.kornia_affine_theta(io.kornia_rotation_matrix2d(center, torch.tensor([deg])), (H, W), (H, W))
        grid = F.affine_grid(theta, [1, 1, H, W], align_corners=True)
        t = theta[0].numpy().astype(f32)
        xs = np.repeat(torch.linspace(-1, 1, W).numpy()[None, :], H, 0)
        ys = np.repeat(torch.linspace(-1, 1, H).numpy()[:, None], W, 1)
        g = grid[0].numpy()
        cands = {"fma(t1,y,t0*x)+t2": fma(t[0, 1], ys, (t[0, 0] * xs).astype(f32)) + t[0, 2],
                 "fma(t0,x,t1*y)+t2": fma(t[0, 0], xs, (t[0, 1] * ys).astype(f32)) + t[0, 2],
                 "(x*t0+y*t1)+t2": ((xs * t[0, 0]).astype(f32) + (ys * t[0, 1]).astype(f32)).astype(f32) + t[0, 2]}
        print(f"affine_grid {H}x{W} {deg} deg, x coordinate differs in:", {k: int((v.astype(f32) != g[..., 0]).sum()) for k, v in cands.items()}, "of", H * W)
        out = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0, 0].numpy()
        ix = ((g[..., 0] + f32(1)) * f32((W - 1) / 2)).astype(f32)
        iy = ((g[..., 1] + f32(1)) * f32((H - 1) / 2)).astype(f32)
        xf, yf = np.floor(ix), np.floor(iy)
        wx, wy = (ix - xf).astype(f32), (iy - yf).astype(f32)
        ex, sy = (f32(1) - wx).astype(f32), (f32(1) - wy).astype(f32)
        nw, ne, sw, se = (sy * ex).astype(f32), (sy * wx).astype(f32), (wy * ex).astype(f32), (wy * wx).astype(f32)
        im = img[0, 0].numpy()

        def val(yy, xx):
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            return np.where(ok, im[np.clip(yy, 0, H - 1).astype(int), np.clip(xx, 0, W - 1).astype(int)], f32(0)).astype(f32)

        vnw, vne, vsw, vse = val(yf, xf), val(yf, xf + 1), val(yf + 1, xf), val(yf + 1, xf + 1)
        c = {"mul + 3 fma (nw, ne, sw, se)": fma(vse, se, fma(vsw, sw, fma(vne, ne, (vnw * nw).astype(f32)))),
             "4 mul + 3 add": (((vnw * nw).astype(f32) + (vne * ne).astype(f32)).astype(f32) + (vsw * sw).astype(f32)).astype(f32) + (vse * se).astype(f32)}
        print(f"grid_sample {H}x{W} {deg} deg differs in:", {k: int((v.astype(f32) != out).sum()) for k, v in c.items()}, "of", H * W)


if __name__ == "__main__":
    main()
