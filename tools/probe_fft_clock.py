"""Shader-cycle stamps of the phases of one block of the fused inverse FFT kernel (debug build only).

  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC -Iinclude -DEQA_FFT_CLOCK equiadapt_amd/csrc/*.hip -o build_variants/libeqa_fftclock.so
  EQA_LIB=$PWD/build_variants/libeqa_fftclock.so python tools/probe_fft_clock.py

Thread 0 of the middle block of the grid stamps the phases of the fused inverse (loads issued and returned, column
transform with its LDS writes, wait at the barrier, row transform with the epilogue, window-sum pieces) and of the fused
forward transform (loads, row transform + LDS writes, barrier, LDS reads + column transform, stores issued).
Numbers: HISTORY.md section 3.4 (the inverse kernel) and section 6.
"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiadapt_amd import _lib
from equiadapt_amd.images.canonicalization_networks import fftconv
dev = torch.device("cuda:0")
x = torch.randn(256, 256, 92, 92, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(256, 256, 5, 5, device=dev) / 80
b = torch.randn(256, device=dev)
B = fftconv.filter_spectra(w)
raw = ctypes.CDLL(_lib.SO_PATH)
if not hasattr(raw, "eqa_debug_fft_clock"):
    sys.exit("this library was built without -DEQA_FFT_CLOCK")
names = ["loads issued and returned", "column transform + LDS writes", "wait at the barrier", "row transform + epilogue",
         "window-sum pieces (2 barriers)"]
out = (ctypes.c_ulonglong * 32)()
for rep in range(3):
    for _ in range(5):
        fftconv.conv5x5(x, B, b, True, b, True, sums_k=5)
    torch.cuda.synchronize()
    assert raw.eqa_debug_fft_clock(out) == 0
    n = len(names)
    tot = sum(out[:n]) or 1    # 0: the pipelined kernel ran instead (stamps [16..29])
    print("inverse: " + " | ".join(f"{n}: {v} ({100 * v / tot:.0f} %)" for n, v in zip(names, out[:n])), f"| total {tot} cycles")
    fn = ["loads issued and returned", "row transform + LDS writes", "wait at the barrier", "LDS reads + column transform", "stores issued"]
    ft = sum(out[8:13])
    print("forward: " + " | ".join(f"{n}: {v} ({100 * v / ft:.0f} %)" for n, v in zip(fn, out[8:13])), f"| total {ft} cycles")
    pn = ["wait for the loads", "column transform + LDS writes", "wait at A", "issue next loads", "wait at B"]
    cn = ["wait at A", "passes up to the last LDS read", "wait at B", "last pass + pieces"]
    if sum(out[16:21]):
        items = 64
        print("pipeline producer (cycles per item): " + " | ".join(f"{n}: {v // items}" for n, v in zip(pn, out[16:21])), f"| total {sum(out[16:21]) // items}")
        print("pipeline consumer (cycles per item): " + " | ".join(f"{n}: {v // items}" for n, v in zip(cn, out[24:28])), f"| total {sum(out[24:28]) // items}")
