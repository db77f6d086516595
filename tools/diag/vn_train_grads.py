"""Parameter gradients of VNSmall's training step: product fast path vs op-by-op fp32 / fp64 (GPU) vs the reference golden."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import equiadapt_amd as ea  # noqa: E402

dev = torch.device("cuda:0")
g = torch.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "pointcloud_n1024.pt"), weights_only=False)
for grp in ("n1024", "k16"):
    t = g[grp]["mean_train"]
    hp = types.SimpleNamespace(n_knn=t["k"], pooling="mean")

    def run(fast, dtype):
        net = ea.VNSmall(hp)
        net.load_state_dict(t["state"])
        net.dropout.p = 0.0
        net = net.to(dev).to(dtype).train()
        os.environ["EQA_TRAIN_FAST"] = fast
        out = net(t["x"].to(dev).to(dtype))
        (out * t["w"].to(dev).to(dtype)).sum().backward()
        os.environ.pop("EQA_TRAIN_FAST")
        return {n: p.grad.double().cpu() for n, p in net.named_parameters() if p.grad is not None}

    fast, op32, op64 = run("1", torch.float32), run("0", torch.float32), run("0", torch.float64)
    for n in op64:
        s = op64[n].abs().max().item() + 1e-30
        d = lambda a: (a[n] - op64[n]).abs().max().item() / s if n in a else float("nan")  # noqa: E731
        gold = {k: v.double() for k, v in t["grads"].items()}
        print(f"{grp} {n:40s} scale {s:9.3e}  rel.err vs fp64: fast {d(fast):.2e}  op32 {d(op32):.2e}  golden {d(gold):.2e}")
