"""The channel contraction of the FFT convolution three ways on the same operands: the fp32 matrix instruction (eqa_fft48k5_cgemm3m),
and the bf16 matrix cores on exact three-piece splits with 9 / 6 piece products (eqa_fft48k5_cgemm3m_bf16x3); time per launch and
the distance of each result to an fp64 evaluation of the same products.

    python tools/kbench_gemm_pieces.py [--m 1024] [--cin 256] [--cout 256]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiadapt_amd import _lib                                                     # noqa: E402
from equiadapt_amd.images.canonicalization_networks import fftconv as fc          # noqa: E402


def measure(M: int = 1024, Cin: int = 256, Cout: int = 256, reps: int = 10, dev=None) -> dict:
    """{mode: {"ms", "tflops_fp32_equivalent", "max_err_vs_fp64", "rms_err_vs_fp64"}} for mode in f32 / 9 / 6, + "scale"."""
    dev = dev or torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    bank = (torch.randn(Cout, Cin, 5, 5, generator=g) / (5.0 * Cin ** 0.5)).to(dev)
    B = fc.filter_spectra3m(bank)
    V = fc.spectra_buffer(M, 2 * Cin, dev)
    V.normal_(generator=torch.Generator(device=dev).manual_seed(1))
    st = torch.cuda.current_stream().cuda_stream
    vb = V.abs().max().reshape(1)            # the bound the fp16 form ("h3") asks for

    def run(mode):
        Mo = fc.spectra_buffer(M, 2 * Cout, dev)
        if mode == "f32":
            _lib.check(lib.eqa_fft48k5_cgemm3m(V.data_ptr(), B.data.data_ptr(), Mo.data_ptr(), M, Cin, Cout, st), "cgemm3m")
        elif mode == "h3":
            bh, b_scale = B.pieces_f16()
            _lib.check(lib.eqa_fft48k5_cgemm3m_f16x2(V.data_ptr(), bh.data_ptr(), Mo.data_ptr(), M, Cin, Cout, vb.data_ptr(), 1, b_scale, st), "f16x2")
        else:
            _lib.check(lib.eqa_fft48k5_cgemm3m_bf16x3(V.data_ptr(), B.pieces().data_ptr(), Mo.data_ptr(), M, Cin, Cout, int(mode[0]), st), "bf16x3")
        return Mo

    flops = 3.0 * 2 * fc.F * M * Cin * Cout
    fs = [0, 1, 577, fc.F - 1]
    want = torch.bmm(V[fs, :M].double(), fc.filter_spectra(bank)[fs].double())     # fp64 truth on a sample of frequencies
    out = {"scale": want.abs().max().item(), "shape": f"{fc.F} x [{M} x {Cin}].[{Cin} x {Cout}] complex"}
    for mode in ("f32", "9", "6", "9w", "6w") + (("h3",) if fc.f16_form_takes(Cin, Cout) else ()):
        # "9" / "6": the form the library chooses (the block form when Cout % 256 == 0); "9w" / "6w": the wave form forced (option 2)
        _lib.check(lib.eqa_set_option(2, 1 if mode.endswith("w") else 0), "set_option")
        for _ in range(3):
            Mo = run(mode)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            Mo = run(mode)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        d = (Mo[fs, :M].double() - want).abs()
        _lib.check(lib.eqa_set_option(2, 0), "set_option")
        out[mode] = {"ms": ms, "tflops_fp32_equivalent": flops / ms * 1e-9, "max_err_vs_fp64": d.max().item(),
                     "rms_err_vs_fp64": d.pow(2).mean().sqrt().item()}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1024)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    r = measure(args.m, args.cin, args.cout, args.reps)
    print(r["shape"], " max |fp64 result| on the sampled frequencies:", f"{r['scale']:.3e}")
    for mode in ("f32", "9", "6", "9w", "6w") + (("h3",) if "h3" in r else ()):
        m = r[mode]
        what = ("fp32 matrix instruction" if mode == "f32" else "two fp16 pieces, 3 products" if mode == "h3"
                else f"bf16 pieces, {mode[0]} products" + (" (wave form)" if mode.endswith("w") else ""))
        print(f"{what:>38}: {m['ms']:7.3f} ms  {m['tflops_fp32_equivalent']:6.1f} TFLOP/s fp32-equivalent   |result - fp64| max {m['max_err_vs_fp64']:.3e} rms {m['rms_err_vs_fp64']:.3e}")


if __name__ == "__main__":
    main()
