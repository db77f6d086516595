"""Micro-benchmark of the Winograd F(2x2,5x5) pipeline stages at the headline layer shape (64 x 92 x 92 x 256),
F(m x m, 5x5) with m = 2 or 4:
python tools/kbench_wino.py [--reps 20].  Variant builds: EQA_LIB=build_variants/libeqa_X.so."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiadapt_amd import _lib  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--hw", type=int, default=92)
    ap.add_argument("--c", type=int, default=256)
    ap.add_argument("--m", type=int, default=4, help="output tile size (2 or 4)")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    n, H, C = args.n, args.hw, args.c
    OH = H - 4
    m = args.m
    P = (m + 4) ** 2
    t = n * (OH // m) ** 2
    f_in, f_out, f_sums = (getattr(lib, f"eqa_winograd_f{m}k5_{k}") for k in ("input", "output", "output_sums"))
    x = torch.randn(n, H, H, C, device=dev)
    V = torch.empty(P * t * C, device=dev)
    M = torch.randn(P * t * C, device=dev)
    U = torch.randn(P, C, C, device=dev) * 0.05
    y = torch.empty(n, OH, OH, C, device=dev)
    bias = torch.randn(C, device=dev)
    S = torch.empty(n, C, 5, 5, dtype=torch.float64, device=dev)
    ws = torch.empty(max(lib.eqa_winograd_f2k5_output_sums_workspace_bytes(n, OH, C, 5), 4) // 4, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    Vb, Mb = V.view(t, P, C).permute(1, 0, 2), M.view(t, P, C).permute(1, 0, 2)
    gb_v = P * t * C * 4 / 1e9
    gb_x = x.numel() * 4 / 1e9
    ms = timeit(lambda: _lib.check(f_in(x.data_ptr(), V.data_ptr(), bias.data_ptr(), 1, n, H, H, C, st), "in"), args.reps)
    print(f"input transform   {ms*1e3:8.1f} us  {(gb_v + gb_x)/ms*1e3:7.0f} GB/s  (write {gb_v:.2f} GB + read {gb_x:.2f} GB)")
    ms = timeit(lambda: torch.bmm(Vb, U, out=Mb), args.reps)
    print(f"batched GEMM (library)  {ms*1e3:8.1f} us  {2*P*t*C*C/ms/1e9:7.1f} TFLOP/s")
    from equiadapt_amd import ops
    if ops.plane_gemm_supported(C, C):
        Upk = ops.pack_plane_gemm_weights(U)
        V3, M3 = V.view(t, P, C), M.view(t, P, C)
        ms = timeit(lambda: ops.plane_gemm(V3, Upk, M3, t), args.reps)
        print(f"eqa_plane_gemm          {ms*1e3:8.1f} us  {2*P*t*C*C/ms/1e9:7.1f} TFLOP/s")
    ms = timeit(lambda: _lib.check(f_out(M.data_ptr(), bias.data_ptr(), 1, y.data_ptr(), n, OH, OH, C, st), "out"), args.reps)
    print(f"output transform  {ms*1e3:8.1f} us  {(gb_v + y.numel()*4/1e9)/ms*1e3:7.0f} GB/s")
    ms = timeit(lambda: _lib.check(f_sums(M.data_ptr(), bias.data_ptr(), 1, S.data_ptr(), ws.data_ptr(), n, OH, OH, C, 5, st), "sums"), args.reps)
    print(f"output + sums     {ms*1e3:8.1f} us  {gb_v/ms*1e3:7.0f} GB/s")


def lift():
    from equiadapt_amd import ops
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    x = torch.randn(256, 3, 96, 96, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(256, 3, 5, 5, device=dev) / 9).contiguous(memory_format=torch.channels_last)
    b = torch.randn(256, device=dev)
    wpk = ops.pack_lift_weights(w)
    gf = 2 * 256 * 92 * 92 * 256 * 75 / 1e9
    ms = timeit(lambda: ops.lift_conv_nhwc(x, wpk, b, True, 5, 5), 20)
    print(f"lift conv (MFMA)  {ms*1e3:8.1f} us  {gf/ms:7.1f} TFLOP/s (75-tap)  {256*92*92*256*4/ms/1e6:7.0f} GB/s written")
    ms = timeit(lambda: F.conv2d(x, w), 20)
    print(f"lift conv (MIOpen){ms*1e3:8.1f} us  {gf/ms:7.1f} TFLOP/s")


def fft():
    """The FFT convolution's stages at the headline shape (256 x 92 x 92 x 256 -> 88 x 88 x 256): forward transform, batched GEMM,
    inverse transform + window sums; and the training-side pieces."""
    from equiadapt_amd.images.canonicalization_networks import fftconv
    lib = _lib.load()
    dev = torch.device("cuda:0")
    n, C, H = int(os.environ.get("KB_N", 256)), 256, 92
    x = torch.randn(n, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, C, 5, 5, device=dev) / 80
    bias = torch.randn(C, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    OH = H - 4
    M = n * 4
    T = torch.empty(max(lib.eqa_fft48k5_workspace_bytes(n, H, OH, C), 4) // 4, device=dev)
    V = fftconv.spectra_buffer(M, 2 * C, dev)
    spectra = V.numel() * 4 / 1e9
    ms = timeit(lambda: _lib.check(lib.eqa_fft48k5_input(x.data_ptr(), T.data_ptr(), V.data_ptr(), bias.data_ptr(), 1, n, H, H, C, st), "in"), 10)
    print(f"fft forward transform   {ms*1e3:8.1f} us  {(x.numel()*4/1e9 + spectra)/ms*1e3:7.0f} GB/s  (read {x.numel()*4/1e9:.2f} GB + write {spectra:.2f} GB)")
    ms = timeit(lambda: fftconv.filter_spectra(w), 5)
    print(f"filter spectra          {ms*1e3:8.1f} us")
    B = fftconv.filter_spectra(w)
    Mo = fftconv.spectra_buffer(M, 2 * C, dev)
    ms = timeit(lambda: torch.bmm(V, B, out=Mo), 10)
    print(f"batched GEMM            {ms*1e3:8.1f} us  {2*fftconv.F*M*512*512/ms/1e9:7.1f} TFLOP/s")
    T2 = torch.empty(max(lib.eqa_fft48k5_workspace_bytes(n, OH, OH, C), 4) // 4, device=dev)
    S = torch.empty(n, C, 5, 5, dtype=torch.float64, device=dev)
    ws = torch.empty(n * OH * 2 * C * 9, device=dev)
    ms = timeit(lambda: _lib.check(lib.eqa_fft48k5_output_sums(Mo.data_ptr(), T2.data_ptr(), bias.data_ptr(), 1, S.data_ptr(), ws.data_ptr(), n, OH, OH, C, 5, st), "out"), 10)
    print(f"fft inverse + sums      {ms*1e3:8.1f} us  {spectra/ms*1e3:7.0f} GB/s")
    y = torch.empty(n, C, OH, OH, device=dev).contiguous(memory_format=torch.channels_last)
    ms = timeit(lambda: _lib.check(lib.eqa_fft48k5_output(Mo.data_ptr(), T2.data_ptr(), bias.data_ptr(), 1, y.data_ptr(), n, OH, OH, C, st), "out"), 10)
    print(f"fft inverse (map)       {ms*1e3:8.1f} us  {(spectra + y.numel()*4/1e9)/ms*1e3:7.0f} GB/s")
    del T, T2, ws
    dy = torch.randn(n, C, OH, OH, device=dev).contiguous(memory_format=torch.channels_last)
    ms = timeit(lambda: fftconv.grad_spectra(dy), 5)
    print(f"gradient-tile spectra   {ms*1e3:8.1f} us")
    G = fftconv.grad_spectra(dy)
    ms = timeit(lambda: fftconv.filter_grad(V, dy, C, G), 5)
    print(f"filter gradient         {ms*1e3:8.1f} us  (GEMM over the tiles + inverse on the 5x5 support)")
    ms = timeit(lambda: fftconv.input_grad(dy, w, G), 5)
    print(f"input gradient          {ms*1e3:8.1f} us  (spectra of the swapped bank + GEMM + overlap-add inverse)")


if __name__ == "__main__":
    if "--fft" in sys.argv:
        fft()
        sys.exit(0)
    if "--lift" in sys.argv:
        sys.argv.remove("--lift")
        lift()
        sys.exit(0)
    main()
