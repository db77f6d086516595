"""Per-wave cycle stamps of the lifting convolution (debug build only).

  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -shared -fPIC -Iinclude -DEQA_LIFT_CLOCK equiadapt_amd/csrc/*.hip -o build_variants/libeqa_clock.so
  EQA_LIB=$PWD/build_variants/libeqa_clock.so python tools/probe_lift_clock.py

Prints shader cycles per tile (5120 = the tile's MFMAs alone), the clock each wave saw (s_memtime / s_memrealtime), the
spread of start / end times and the host-measured launch period.  Numbers: HISTORY.md section 3.4.
"""
import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiadapt_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
x = torch.randn(256, 3, 96, 96, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(256, 3, 5, 5, device=dev) / 9; b = torch.randn(256, device=dev)
wpk = ops.pack_lift_weights(w)
raw = ctypes.CDLL(_lib.SO_PATH)
out = (ctypes.c_ulonglong * (4 * 2048))()
for rep in range(2):
    for _ in range(20): ops.lift_conv_nhwc(x, wpk, b, True, 5, 5)
    torch.cuda.synchronize()
    assert raw.eqa_debug_lift_clock(out) == 0
    a = np.array(list(out), dtype=np.uint64).reshape(2048, 4); a = a[a[:, 2] > 0]
    cyc = a[:, 0].astype(np.float64); t0 = a[:, 1].astype(np.float64); t1 = a[:, 2].astype(np.float64)
    xcc = (a[:, 3] >> np.uint64(32)).astype(np.int64) & 0xf; hw = a[:, 3].astype(np.int64) & 0xffffffff
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    base = t0.min()
    st = (t0 - base) / 100; en = (t1 - base) / 100; dur = en - st
    print(f"start  min {st.min():.1f} med {np.median(st):.1f} max {st.max():.1f} us | end min {en.min():.1f} med {np.median(en):.1f} max {en.max():.1f} us | dur min {dur.min():.1f} med {np.median(dur):.1f} max {dur.max():.1f}")
    print(f"  waves {len(a)}  cycles/tile med {np.median(cyc)/ (70656*4/len(a)):.0f} (MFMA need 5120)")
    print("  GHz min/med/max", (cyc / dur / 1000).min().round(3), np.median(cyc / dur / 1000).round(3), (cyc / dur / 1000).max().round(3))
    for xc in range(0):
        m = xcc == xc
        if m.any(): print(f"  xcc {xc}: waves {m.sum()} dur med {np.median(dur[m]):.1f} max {dur[m].max():.1f} end max {en[m].max():.1f}  distinct (se,sh,cu) {len(set(zip(se[m], sh[m], cu[m])))}")
    late = np.argsort(-en)[:8]
    print("  latest waves:", [(int(i), round(float(st[i]), 1), round(float(en[i]), 1), int(xcc[i]), int(se[i]), int(cu[i])) for i in late])

import time
hist = (ctypes.c_ulonglong * 64)()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): ops.lift_conv_nhwc(x, wpk, b, True, 5, 5)
e1.record(); torch.cuda.synchronize()
assert raw.eqa_debug_lift_hist(hist) == 0
h = np.sort(np.array(list(hist), dtype=np.uint64).astype(np.float64))
d = np.diff(h)[-30:]
print(f"host: {e0.elapsed_time(e1)/40*1000:.1f} us per launch; device: wave-0 start to next launch's wave-0 start = {np.median(d):.0f} ticks (median) -> realtime counter {np.median(d)/(e0.elapsed_time(e1)/40*1000):.2f} MHz")
