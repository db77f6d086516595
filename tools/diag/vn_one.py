"""One eqa_vnsmall_fwd launch series for counter collection: python tools/diag/vn_one.py [B]"""
import os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import equiadapt_amd as ea
from equiadapt_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
net = ea.VNSmall(types.SimpleNamespace(n_knn=20, pooling="mean")).to(dev).eval()
x = torch.randn(B, 3, 1024, device=dev)
for _ in range(3):
    ops.vnsmall_forward(x, net.packed_parameters(), 20, "mean")
torch.cuda.synchronize()
