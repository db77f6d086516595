"""Analyse a case dumped by tools/fuzz_r03.py --dump: is the kernel's deviation from the fp64 evaluation one (or a few) max-pool
arg-max flips at near-tied scores?  python tools/diag/vn_fuzz_case.py dump.pt"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import equiadapt_amd as ea
from equiadapt_amd.pointcloud.canonicalization_networks.equivariant_networks import get_graph_feature_cross

d = torch.load(sys.argv[1])
dev = torch.device("cuda")
x, k, pooling, idx = d["x"].to(dev), d["k"], d["pooling"], d["idx"].to(dev)
net = ea.VNSmall(types.SimpleNamespace(n_knn=k, pooling=pooling))
net.load_state_dict(d["state"])
net = net.to(dev).eval()
B, _, N = x.shape
print("case: B", B, "N", N, "k", k, pooling)
with torch.no_grad():
    a = net(x)                                   # kernel
    nd = net.double()
    xx = x.double()
    h = nd.conv_pos(get_graph_feature_cross(xx.unsqueeze(1), k, idx.long()))          # (B, C, 3, N, k)
    dd = nd.pool.map_to_dir(h.transpose(1, -1)).transpose(1, -1)
    score = (h * dd).sum(2)                                                              # (B, C, N, k)
    size = (h * dd).abs().sum(2).amax(-1)
    top = score.topk(2, dim=-1)
    margin = (top.values[..., 0] - top.values[..., 1]) / size.clamp_min(1e-30)          # (B, C, N)
    def tail(pooled):
        return nd.conv2(nd.bn1(nd.conv1(pooled))).mean(dim=-1)[:, :3]
    best = top.indices[..., 0]
    def pool_with(choice):
        ix = choice[:, :, None, :, None].expand(-1, -1, 3, -1, 1)
        return torch.gather(h, 4, ix).squeeze(-1)
    truth = tail(pool_with(best))
    print("fp64 recomputed vs dumped op64:", (truth.cpu() - d["op64"]).abs().max().item())
    err = (a.double() - truth)
    print("kernel - fp64: max", err.abs().max().item(), "per cloud", err.abs().amax((1, 2)).tolist())
    # flip the arg-max at the n smallest-margin (cloud, channel, point) sites, one at a time, and see which flips explain the error
    flat = margin.flatten()
    order = flat.argsort()[:12]
    for o in order.tolist():
        b, c, n = o // (margin.shape[1] * N), (o // N) % margin.shape[1], o % N
        ch = best.clone()
        ch[b, c, n] = top.indices[b, c, n, 1]
        out = tail(pool_with(ch))
        print(f"site cloud {b} ch {c} pt {n}: margin {flat[o].item():.3e}  |flip - truth| {(out - truth).abs().max().item():.3e}  "
              f"|kernel - flip| {(a.double() - out).abs().max().item():.3e}")
    net.float()
    s32 = net.conv_pos(get_graph_feature_cross(x.unsqueeze(1), k, idx.long()))
    d32 = net.pool.map_to_dir(s32.transpose(1, -1)).transpose(1, -1)
    sc32 = (s32 * d32).sum(2)
    print("torch fp32 score error / size (max):", ((sc32.double() - score).abs() / size.clamp_min(1e-30)[..., None]).max().item())
