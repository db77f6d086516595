"""Point-cloud path and backward kernels: python tools/kbench_pc.py"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import equiadapt_amd as ea  # noqa: E402
from equiadapt_amd import ops  # noqa: E402
from equiadapt_amd.images.utils import device_tables  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


dev = torch.device("cuda:0")
hp = types.SimpleNamespace(n_knn=20, pooling="mean")
net = ea.VNSmall(hp).to(dev).eval()
can = ea.EquivariantPointcloudCanonicalization(net, hp).to(dev).eval()
for B in (16, 64, 256, 2048):
    x = torch.randn(B, 3, 1024, device=dev)
    with torch.no_grad():
        ms = timeit(lambda: can(x))
    print(f"pointcloud canonicalize (fused VNSmall + GS + rotate) B={B}: {ms:8.2f} ms  {B/ms*1e3:9.0f} clouds/s")
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        can(x)
    print(f"   peak mem {torch.cuda.max_memory_allocated()/1e6:.0f} MB")

B, S = 256, 224
x = torch.randn(B, 3, S, S, device=dev)
gy = torch.randn(B, 3, S, S, device=dev)
gidx = torch.randint(0, 8, (B,), generator=torch.Generator().manual_seed(1)).to(dev, torch.int32)
th_c, fl_c = device_tables("canonicalize", 8, False, (2 * S, 2 * S), dev)
from equiadapt_amd import _lib as _l  # noqa: E402
for opt, label in ((0, "frame gather + fold"), (1, "atomic scatter")):
    _l.load().eqa_set_option(0, opt)
    for want_src, want_angle in ((False, True), (True, False), (True, True)):
        ms = timeit(lambda: ops.group_action_bwd(x, gy, gidx, th_c, fl_c, None, S // 2, (S // 2, S // 2), want_src, want_angle), 20)
        print(f"canonicalize backward [{label}] src={want_src} angle={want_angle}: {ms*1e3:8.1f} us")
_l.load().eqa_set_option(0, 0)

# un-padded input gradient (invert action): deterministic gather; EQA option 0 = 1 forces the atomic scatter
from equiadapt_amd import _lib  # noqa: E402
lib = _lib.load()
th_i, fl_i, cm_i = device_tables("invert", 8, False, (S, S), dev)
for name, opt in (("gather", 0), ("atomic scatter", 1)):
    lib.eqa_set_option(0, opt)
    ms = timeit(lambda: ops.group_action_bwd(x, gy, gidx, th_i, fl_i, None, 0, (0, 0), True, False), 20)
    print(f"invert_action input gradient, {name}: {ms*1e3:8.1f} us")
lib.eqa_set_option(0, 0)
f8 = torch.randn(64, 64, S, S, device=dev)
g8 = torch.randn(64, 64, S, S, device=dev)
gi8 = gidx[:64]
for name, opt in (("gather", 0), ("atomic scatter", 1)):
    lib.eqa_set_option(0, opt)
    ms = timeit(lambda: ops.group_action_bwd(f8, g8, gi8, th_i, fl_i, cm_i, 0, (0, 0), True, False), 10)
    print(f"invert_action input gradient, regular rep 64x64ch, {name}: {ms*1e3:8.1f} us")
lib.eqa_set_option(0, 0)
