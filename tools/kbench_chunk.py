"""Lifting convolution + forward FFT transform of the headline batch (256 x 96 x 96 x 3 -> 92 x 92 x 256 -> spectra), whole batch at once
against chunks of images whose lifted map stays in ONE re-used buffer (does the 256 MiB Infinity Cache keep it from HBM?).
python tools/kbench_chunk.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from equiadapt_amd import _lib, ops
from equiadapt_amd.images.canonicalization_networks import fftconv

dev = torch.device("cuda")
lib = _lib.load()
B, C = 256, 256
torch.manual_seed(0)
x = torch.randn(B, 3, 96, 96, device=dev).contiguous(memory_format=torch.channels_last)
bank = torch.randn(C, 3, 5, 5, device=dev) / 8
bias = torch.randn(C, device=dev)
wpk = ops.pack_lift_weights(bank)
M = B * 4
st = torch.cuda.current_stream().cuda_stream
big = torch.empty(3 * 1024 ** 3 // 4, device=dev)        # a 3 GB scribble between runs: nothing useful survives in the caches


def whole(V):
    g = ops.lift_conv_grouped(x, wpk, bias, True, 5, 5)
    _lib.check(lib.eqa_fft48k5_input_grouped_at(g.data_ptr(), V.data_ptr(), None, 0, B, 92, 92, C, 0, M, st), "at")


def chunked(V, n, scratch):
    for i0 in range(0, B, n):
        nb = min(n, B - i0)
        g = ops.lift_conv_grouped(x[i0:i0 + nb], wpk, bias, True, 5, 5, out=scratch)
        _lib.check(lib.eqa_fft48k5_input_grouped_at(g.data_ptr(), V.data_ptr(), None, 0, nb, 92, 92, C, i0 * 4, M, st), "at")


def timed(fn, reps=10):
    fn()
    ts = []
    for _ in range(reps):
        big.add_(1.0)                                     # evict
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


V0, V1 = fftconv.spectra_buffer(M, 2 * C, dev), fftconv.spectra_buffer(M, 2 * C, dev)
whole(V0)
print("whole batch: %.3f ms" % timed(lambda: whole(V0)))
for n in (8, 12, 16, 20, 24, 28, 32, 64, 128):
    scratch = torch.empty(n * C * 92 * 92, device=dev)
    chunked(V1, n, scratch)
    torch.cuda.synchronize()
    same = torch.equal(V0, V1)
    print("chunks of %3d images (%5.0f MB lifted): %.3f ms   spectra identical: %s" % (n, n * C * 92 * 92 * 4 / 2 ** 20, timed(lambda: chunked(V1, n, scratch)), same))
    V1.zero_()
