"""How far apart are two training steps of the canonicalization network that differ only in where the batch statistics are summed
(convolution epilogues vs eqa_bn_stats_nhwc's pass)?  Prints, per parameter, the gradient difference relative to the gradient's
maximum for (epilogue vs epilogue), (pass vs pass), (epilogue vs pass), and the same against an fp64 evaluation on the op path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import equiadapt_amd as ea  # noqa: E402

dev = torch.device("cuda:0")


def run(flag, double=False):
    os.environ["EQA_TRAIN_EPILOGUE_STATS"] = flag
    torch.manual_seed(3)
    net = ea.ESCNNEquivariantNetwork((3, 100, 100), 64 // 4, 5, "rotation", 4, 4).to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0 if double else m.p
    x = torch.randn(12, 3, 100, 100, device=dev)
    if double:
        net, x = net.double(), x.double()
    torch.manual_seed(11)
    out = net(x)
    out.square().sum().backward()
    return out.detach(), [p.grad.clone() for p in net.parameters()], [n for n, _ in net.named_parameters()]


a = run("1"); b = run("1"); c = run("0"); d = run("0")
for name, (u, v) in {"epilogue vs epilogue": (a, b), "pass vs pass": (c, d), "epilogue vs pass": (a, c)}.items():
    print(name, "output", ((u[0] - v[0]).abs().max() / v[0].abs().max()).item())
    for gu, gv, n in zip(u[1], v[1], u[2]):
        print(f"   {n:40s} {((gu - gv).abs().max() / gv.abs().max().clamp_min(1e-30)).item():.3e}   |g| max {gv.abs().max().item():.3e}")
