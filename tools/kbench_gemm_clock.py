"""The shader clock the bf16-piece contraction (block form) sustains, and how its cycles divide.

A variant build of the library (-DEQA_BLK_CLOCK=1: first / last s_memtime and the constant 100 MHz counter of one wave; =2: an
s_memtime at every region boundary of a K-stage as well) exports eqa_debug_blk_clock.  The product library has none of this.

    python -c "from equiadapt_amd import _lib; _lib.build(extra_flags=['-DEQA_BLK_CLOCK=1'], out='build_variants/libeqa_clock.so')"
    EQA_LIB=build_variants/libeqa_clock.so python tools/kbench_gemm_clock.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiadapt_amd import _lib                                                     # noqa: E402
from equiadapt_amd.images.canonicalization_networks import fftconv as fc          # noqa: E402


def main():
    lib = _lib.load()
    raw = ctypes.CDLL(os.environ["EQA_LIB"])
    out = (ctypes.c_longlong * 10)()
    dev = torch.device("cuda:0")
    M, Cin, Cout = 1024, 256, 256
    S = Cin // 16
    tiles = -(-(-(-fc.F // 8)) * (-(-M // 128)) * (Cout // 128) // 32)                # block tiles of the busiest block
    g = torch.Generator().manual_seed(0)
    bank = (torch.randn(Cout, Cin, 5, 5, generator=g) / (5.0 * Cin ** 0.5)).to(dev)
    B = fc.filter_spectra3m(bank)
    Mo = fc.spectra_buffer(M, 2 * Cout, dev)
    st = torch.cuda.current_stream().cuda_stream
    for what in ("random operands", "A = 0"):
        V = fc.spectra_buffer(M, 2 * Cin, dev)
        V.normal_() if what == "random operands" else V.zero_()
        vb = torch.ones(1, device=dev) * 8.0
        for terms in (9, 6, 3):
            def run():
                if terms == 3:        # the fp16 form: two pieces, three products
                    bh, b_scale = B.pieces_f16()
                    _lib.check(lib.eqa_fft48k5_cgemm3m_f16x2(V.data_ptr(), bh.data_ptr(), Mo.data_ptr(), M, Cin, Cout, vb.data_ptr(), 1, b_scale, st), "f16x2")
                    return
                _lib.check(lib.eqa_fft48k5_cgemm3m_bf16x3(V.data_ptr(), B.pieces().data_ptr(), Mo.data_ptr(), M, Cin, Cout, terms, st), "bf16x3")
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                run()
            b.record()
            torch.cuda.synchronize()
            raw.eqa_debug_blk_clock(out)
            stages = tiles * S
            mfma = {9: 27, 6: 18, 3: 9}[terms] * 4 * 32
            print(f"{what:>16}, {terms} products: {a.elapsed_time(b) / 10:6.3f} ms per launch; block 100: {out[0]} cycles = {out[0] / stages:6.0f} per K-stage "
                  f"(matrix instructions {mfma}: {mfma * stages / out[0]:.2f} of the cycles), {out[0] / (out[1] * 10.0):5.3f} GHz, "
                  f"tile epilogue {out[2] / tiles:5.0f} cycles" + (f", regions per stage {[round(out[3 + i] / stages) for i in range(5)]}" if out[3] else ""))


if __name__ == "__main__":
    main()
