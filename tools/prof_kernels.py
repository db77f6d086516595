"""Tiny driver for rocprofv3: a few launches of each hot kernel at the headline shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equiadapt_amd import ops  # noqa: E402
from equiadapt_amd.images.utils import device_tables  # noqa: E402

B, S = 256, 224
dev = torch.device("cuda:0")
x = torch.randn(B, 3, S, S, device=dev)
gidx = torch.randint(0, 8, (B,), generator=torch.Generator().manual_seed(1)).to(dev, torch.int32)
th_c, fl_c = device_tables("canonicalize", 8, False, (2 * S, 2 * S), dev)
th_i, fl_i, cm_i = device_tables("invert", 8, False, (S, S), dev)
fm = torch.randn(B, 32, 8, 84, 84, device=dev)
for _ in range(int(os.environ.get("REPS", "5"))):
    ops.canon_transform(x, gidx, th_c, fl_c, S // 2)
    ops.invert_action(x, gidx, th_i, fl_i, None)
    ops.group_pool_argmax(fm)
torch.cuda.synchronize()
