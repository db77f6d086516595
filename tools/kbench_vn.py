"""VNSmall forward kernels in isolation (HIP events): python tools/kbench_vn.py
  * eqa_vnsmall_fwd, four lanes per point vs one thread per point (eqa_set_option key 1), B in {16, 64, 256, 2048}, N = 1024
  * other neighbourhood sizes (k = 8, 16, 32), max pooling
  * eqa_vn_knn (the training path's neighbour kernel)"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import equiadapt_amd as ea  # noqa: E402
from equiadapt_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")


def ev_time(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for pooling in ("mean", "max"):
    for k in (20, 16, 8, 32):
        net = ea.VNSmall(types.SimpleNamespace(n_knn=k, pooling=pooling)).to(dev).eval()
        prm = net.packed_parameters()
        for B in ((16, 64, 256, 2048) if k == 20 else (64, 2048)):
            x = torch.randn(B, 3, 1024, device=dev)
            line = f"vnsmall_fwd {pooling:4s} k={k:2d} B={B:4d}:"
            for choice, name in ((2, "quad"), (1, "single")):
                if choice == 1 and k != 20:
                    continue
                lib.eqa_set_option(1, choice)
                ms = ev_time(lambda: ops.vnsmall_forward(x, prm, k, pooling))
                line += f"  {name} {ms*1e3:8.1f} us = {B/ms*1e3/1e3:7.1f} k clouds/s"
            lib.eqa_set_option(1, 0)
            print(line, flush=True)

for k in (20, 32):
    for B in (64, 2048):
        x = torch.randn(B, 3, 1024, device=dev)
        idx = torch.empty(B, 1024, k, dtype=torch.int32, device=dev)
        ms = ev_time(lambda: lib.eqa_vn_knn(x.data_ptr(), idx.data_ptr(), B, 1024, k, None))
        print(f"vn_knn k={k} B={B}: {ms*1e3:8.1f} us")
