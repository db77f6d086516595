"""Distance to an fp64 evaluation of the three forms of the channel contraction (fp32 matrix instruction; bf16 pieces with 9 / 6
piece products) on the shapes of the parity test, two seeds each: max and rms over all frequencies.

    python tools/kbench_gemm_error.py
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiadapt_amd import _lib
from equiadapt_amd.images.canonicalization_networks import fftconv
lib = _lib.load(); dev = torch.device("cuda:0"); fftconv.GEMM_PIECES = "f32"   # `contract` below = the fp32 matrix instruction
for (M, Cin, Cout) in [(36, 64, 128), (36, 64, 64), (130, 32, 128), (64, 256, 256), (257, 96, 192), (1024, 128, 64), (1, 32, 128), (129, 64, 256), (300, 160, 384), (512, 256, 256)]:
    for seed in (0, 1):
        g = torch.Generator().manual_seed(M + Cin + 1 + seed)
        bank = (torch.randn(Cout, Cin, 5, 5, generator=g) / (5.0 * Cin ** 0.5)).to(dev)
        B = fftconv.filter_spectra(bank); B3 = fftconv.filter_spectra3m(bank)
        V = fftconv.spectra_buffer(M, 2 * Cin, dev)
        V.copy_(torch.randn(fftconv.F, M, 2 * Cin, generator=g).to(dev))
        pitch = lib.eqa_fft48k5_tile_pitch(M)
        want = torch.bmm(V.double(), B.double())
        st = torch.cuda.current_stream().cuda_stream
        ref = fftconv.contract(V, B3, M)
        d = (ref.double() - want)[:, :M] if ref.shape[1] != M else (ref.double() - want)
        out = {"f32": (d.abs().max().item(), d.pow(2).mean().sqrt().item())}
        for terms in (9, 6):
            full = torch.zeros((fftconv.F, pitch, 2 * Cout), dtype=torch.float32, device=dev)
            _lib.check(lib.eqa_fft48k5_cgemm3m_bf16x3(V.data_ptr(), B3.pieces().data_ptr(), full.data_ptr(), M, Cin, Cout, terms, st), "x")
            d = full[:, :M].double() - want[:, :M]
            out[str(terms)] = (d.abs().max().item(), d.pow(2).mean().sqrt().item())
        if fftconv.f16_form_takes(Cin, Cout):
            bh, b_scale = B3.pieces_f16()
            vb = V.abs().max().reshape(1)
            full = torch.zeros((fftconv.F, pitch, 2 * Cout), dtype=torch.float32, device=dev)
            _lib.check(lib.eqa_fft48k5_cgemm3m_f16x2(V.data_ptr(), bh.data_ptr(), full.data_ptr(), M, Cin, Cout, vb.data_ptr(), 1, b_scale, st), "h3")
            d = full[:, :M].double() - want[:, :M]
            out["h3"] = (d.abs().max().item(), d.pow(2).mean().sqrt().item())
        print((M, Cin, Cout), seed, " ".join(f"{k}: max {v[0]:.3e} rms {v[1]:.3e}" for k, v in out.items()),
              "9<=f32:", out["9"][0] <= out["f32"][0], out["9"][1] <= out["f32"][1], "6<=f32:", out["6"][0] <= out["f32"][0], out["6"][1] <= out["f32"][1],
              *(("h3<=f32:", out["h3"][0] <= out["f32"][0], out["h3"][1] <= out["f32"][1]) if "h3" in out else ()))
