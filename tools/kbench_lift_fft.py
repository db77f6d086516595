"""The fused lifting convolution + forward FFT-48 transform (eqa_lift5_fft48k5_input, csrc/lift_fft.hip) against the two kernels it
replaces (eqa_lift_conv_grouped then eqa_fft48k5_input_grouped) at the headline shape, back to back, per launch.

    python tools/kbench_lift_fft.py [--batch 256] [--channels 256]
With a library built with -DEQA_LF_CLOCK (EQA_LIB=...): also the per-phase shader cycles of block 0.
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiadapt_amd import _lib, ops                                                   # noqa: E402
from equiadapt_amd.images.canonicalization_networks import fftconv as fc             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    B, C, S = a.batch, a.channels, a.size
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 3, S, S, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    bank = (torch.randn(C, 3, 5, 5, generator=g) / 75 ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(C, generator=g).to(dev)
    wpk = ops.pack_lift_weights(bank)
    H1 = S - 4
    M = B * fc.tiles(H1) ** 2
    st = torch.cuda.current_stream().cuda_stream
    V = fc.spectra_buffer(M, 2 * C, dev)
    T = torch.empty(max(lib.eqa_fft48k5_workspace_bytes(B, H1, H1 - 4, C), 4) // 4, dtype=torch.float32, device=dev)

    def fused():
        _lib.check(lib.eqa_lift5_fft48k5_input(x.data_ptr(), bank.data_ptr(), bias.data_ptr(), 1, V.data_ptr(), B, S, S, C, st), "fused")

    from equiadapt_amd.images.canonicalization_networks.fftconv import LiftedInput
    wp = LiftedInput(x, bank, bias, True).pieces()

    def fused_p():
        _lib.check(lib.eqa_lift5_fft48k5_input_bf16x3(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), 1, V.data_ptr(), B, S, S, C, st), "fused bf16x3")

    wh, w_scale = LiftedInput(x, bank, bias, True).pieces_f16()
    xb = torch.empty(256, dtype=torch.float32, device=dev)
    dcm = torch.empty(256, dtype=torch.float32, device=dev)

    def fused_h():
        _lib.check(lib.eqa_absmax_slots(x.data_ptr(), x.numel(), xb.data_ptr(), st), "absmax")
        _lib.check(lib.eqa_lift5_fft48k5_input_f16x2(x.data_ptr(), wh.data_ptr(), w_scale, xb.data_ptr(), 256, bias.data_ptr(), 1, V.data_ptr(),
                                                     dcm.data_ptr(), B, S, S, C, st), "fused f16x2")

    def two():
        y = ops.lift_conv_grouped(x, wpk, bias, True, 5, 5)
        _lib.check(lib.eqa_fft48k5_input_grouped(y.data_ptr(), T.data_ptr(), V.data_ptr(), None, 0, B, H1, H1, C, st), "grouped")

    def lift_only():
        ops.lift_conv_grouped(x, wpk, bias, True, 5, 5)

    for name, fn in (("fused eqa_lift5_fft48k5_input", fused), ("fused eqa_lift5_fft48k5_input_bf16x3", fused_p),
                     ("eqa_absmax_slots + fused eqa_lift5_fft48k5_input_f16x2", fused_h), ("lift_conv_grouped + fft48k5_input_grouped", two), ("lift_conv_grouped alone", lift_only)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:>56}: {e0.elapsed_time(e1) / a.reps:7.3f} ms per launch  (B = {B}, {C} channels, {S} x {S})")
    raw = ctypes.CDLL(_lib.SO_PATH)
    if hasattr(raw, "eqa_debug_lf_clock"):
        out = (ctypes.c_ulonglong * 32)()
        form = os.environ.get("EQA_LIFT_FFT_FORM")
        (fused_p if form == "bf16x3" else fused_h if form == "h2" else fused)()
        torch.cuda.synchronize()
        assert raw.eqa_debug_lf_clock(out) == 0
        names = ["stage + barrier 1", "prefetch issue", "role work", "barrier 2", "tail row passes", "barrier 3", "column read", "barrier 4"]
        items = max(1, (M * (C // 16) + 255) // 256) if M * (C // 16) >= 256 else 1
        for base, who in ((0, "convolution wave 8"), (8, "column wave 1"), (16, "column wave 0 (packed)")):
            tot = sum(out[base:base + 8])
            print(f"{who}: cycles per item (block 0, {items} items): " + " | ".join(f"{n}: {out[base + i] // items}" for i, n in enumerate(names)) + f" | total {tot // items}")


if __name__ == "__main__":
    main()
