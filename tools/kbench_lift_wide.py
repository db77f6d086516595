"""eqa_lift_conv_wide alone at the reference tutorial's first layer (512 x 3 x 64 x 64, 9 x 9, 64 channels) and at a 224-frame
shape, beside the framework's convolution:   python tools/kbench_lift_wide.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiadapt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, Cin, H, W, K, Cout) in [(512, 3, 64, 64, 9, 64), (512, 3, 64, 64, 7, 64), (256, 3, 96, 96, 9, 256), (512, 1, 64, 64, 5, 64)]:
    x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, K, K, device=dev) / (K * Cin ** 0.5)
    b = torch.randn(Cout, device=dev)
    wpk = ops.pack_lift_weights_wide(w)
    wcl = w.contiguous(memory_format=torch.channels_last)

    def t(fn, reps=10):
        for _ in range(3):
            fn()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e) / reps

    ms = t(lambda: ops.lift_conv_wide(x, wpk, b, True, K, K))
    ms_fw = t(lambda: torch.relu_(F.conv2d(x, wcl, b)))
    fl = 2.0 * B * (H - K + 1) * (W - K + 1) * Cout * Cin * K * K
    print(f"{(B, Cin, H, W, K, Cout)}: eqa_lift_conv_wide {ms:.3f} ms = {fl / ms * 1e-9:.1f} TFLOP/s   framework conv2d + relu {ms_fw:.3f} ms")
