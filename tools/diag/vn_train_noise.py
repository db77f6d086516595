"""How far apart are fp32 evaluations of VNSmall's training-mode forward at B=8 x 1024 points?  Product fast path, op-by-op fp32
path, op-by-op fp64 path (all on the GPU) and the reference-generated golden (CPU fp32)."""
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
        out = net(t["x"].to(dev).to(dtype)).detach().double().cpu()
        os.environ.pop("EQA_TRAIN_FAST")
        return out

    fast, op32, op64, gold = run("1", torch.float32), run("0", torch.float32), run("0", torch.float64), t["vnsmall_out"].double()
    d = lambda a, b: (a - b).abs().max().item()  # noqa: E731
    print(f"{grp}: |out|max {gold.abs().max():.3f}  fast-fp64 {d(fast, op64):.2e}  op32-fp64 {d(op32, op64):.2e}  golden-fp64 {d(gold, op64):.2e}  "
          f"fast-golden {d(fast, gold):.2e}  fast-op32 {d(fast, op32):.2e}")
