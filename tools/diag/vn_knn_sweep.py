"""Where does the quad kNN / fused kernel disagree with torch?  Sweep of (k, N)."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import equiadapt_amd as ea  # noqa: E402
from equiadapt_amd import _lib  # noqa: E402
from equiadapt_amd.pointcloud.canonicalization_networks.equivariant_networks import knn  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for k in (1, 2, 3, 5, 8, 20, 21, 32):
    for N in (32, 40, 47, 48, 64, 70, 100, 513):
        if N < k:
            continue
        x = torch.randn(2, 3, N, device=dev)
        idx = torch.empty(2, N, k, dtype=torch.int32, device=dev)
        assert lib.eqa_vn_knn(x.data_ptr(), idx.data_ptr(), 2, N, k, None) == 0
        torch.cuda.synchronize()
        want = knn(x, k)
        bad = (idx.long().sort(-1).values != want.sort(-1).values).any(-1)
        net = ea.VNSmall(types.SimpleNamespace(n_knn=k, pooling="mean")).to(dev).eval()
        with torch.enable_grad():
            slow = net(x).detach()
        with torch.no_grad():
            fast = net(x)
        e = (fast - slow).abs().max().item()
        flag = "  <<<<" if bad.any() or e > 1e-5 else ""
        print(f"k={k:2d} N={N:4d}: knn mismatching points {int(bad.sum()):4d} (first {bad.nonzero()[:3].tolist()})  fused err {e:.2e}{flag}")
