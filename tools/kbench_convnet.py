"""ConvNetwork's training kernels alone at the tutorial's shapes (2048 views of 3 x 64 x 64; k = 5, 16 / 16 / 32 channels): forward,
filter gradient, data gradient per layer, us per call over a ring of inputs.

    python tools/kbench_convnet.py [--reps 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiadapt_amd import ops                                                     # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=2048)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, k = args.batch, 5
    layers = [(3, 16, 64, 0, True), (16, 16, 30, 0, False), (16, 32, 13, 1, False)]       # Cin, Cout, H = W, pad, planar
    for (cin, cout, H, pad, planar) in layers:
        OH = (H + 2 * pad - k) // 2 + 1
        x = torch.randn(B, cin, H, H, device=dev) if planar else torch.randn(B, H, H, cin, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.1
        wp = ops.pack_conv_s2_weights(w, planar)
        dz = torch.randn(B, OH, OH, cout, device=dev)
        line = f"layer {cin:2d} -> {cout:2d}, {H}x{H} -> {OH}x{OH}:  forward {timed(lambda: ops.conv_s2(x, wp, None, False, cout, k, pad, planar), args.reps):7.1f} us"
        line += f"   filter gradient (+ reduction) {timed(lambda: ops.conv_s2_wgrad(x, dz, k, pad, planar), args.reps):7.1f} us"
        if not planar:
            wd = ops.pack_conv_s2_dgrad_weights(w)
            line += f"   data gradient {timed(lambda: ops.conv_s2_dgrad(dz, wd, (H, H), cin, k, pad), args.reps):7.1f} us"
        print(line)


if __name__ == "__main__":
    main()
