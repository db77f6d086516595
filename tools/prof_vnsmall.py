"""eqa_vnsmall_fwd at B = 64 and B = 2048 (N = 1024, k = 20, mean pooling), a few launches each: the target of tools/collect_sq_counters.sh."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import equiadapt_amd as ea  # noqa: E402
from equiadapt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
net = ea.VNSmall(types.SimpleNamespace(n_knn=20, pooling="mean")).to(dev).eval()
prm = net.packed_parameters()
for B in (64, 2048):
    x = torch.randn(B, 3, 1024, device=dev)
    for _ in range(int(os.environ.get("REPS", "4"))):
        ops.vnsmall_forward(x, prm, 20, "mean")
torch.cuda.synchronize()
