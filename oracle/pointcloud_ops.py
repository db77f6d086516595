"""Oracle for the point-cloud rows of SURVEY.md section 8a (P1-P5).  TEST INFRASTRUCTURE ONLY.

Functional restatement (plain tensors in, plain tensors out) of the reference's VNSmall
canonicalization network, Gram-Schmidt and the SO(3) action, in the reference's unfused op order.
Parameters are passed as a flat ``dict`` whose keys are the reference module's ``state_dict`` names,
so a reference checkpoint / golden fixture feeds it unchanged.

PINNED: checked by tests/test_oracle_golden.py against vectors that tests/golden/make_golden.py
produced by running the unmodified reference sources (gram_schmidt, VN layers directly; VNSmall and
the point-cloud canonicalizer with an annotation-only ``omegaconf`` stand-in -- see that script).
"""
from typing import Dict, Optional

import torch
import torch.nn.functional as F

EPS = 1e-6  # pointcloud/canonicalization_networks/vector_neuron_layers.py:13


def gram_schmidt(vectors: torch.Tensor) -> torch.Tensor:
    """P3: classical Gram-Schmidt on the three rows; no epsilon, no handedness fix (det may be -1).

    common/utils.py:22-51.
    """
    a, b, c = vectors[:, 0], vectors[:, 1], vectors[:, 2]
    e1 = a / torch.norm(a, dim=1, keepdim=True)
    u2 = b - torch.sum(b * e1, dim=1, keepdim=True) * e1
    e2 = u2 / torch.norm(u2, dim=1, keepdim=True)
    u3 = c - torch.sum(c * e1, dim=1, keepdim=True) * e1 - torch.sum(c * e2, dim=1, keepdim=True) * e2
    e3 = u3 / torch.norm(u3, dim=1, keepdim=True)
    return torch.stack([e1, e2, e3], dim=1)


def knn(x: torch.Tensor, k: int) -> torch.Tensor:
    """P1: indices of the k largest -||xi - xj||^2 (self included) -- equivariant_networks.py:15-33.

    x: (B, 3, N) -> (B, N, k) int64.
    """
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x**2, dim=1, keepdim=True)
    neg_sq_dist = -xx - inner - xx.transpose(2, 1)
    return neg_sq_dist.topk(k=k, dim=-1)[1]


def graph_feature_cross(x: torch.Tensor, k: int = 20, idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """P1: (B, 1, 3, N) -> (B, 3, 3, N, k): channels [neighbour - centre, centre, neighbour x centre].

    equivariant_networks.py:36-76.
    """
    B, N = x.size(0), x.size(3)
    x = x.view(B, -1, N)
    if idx is None:
        idx = knn(x, k=k)
    idx = (idx + torch.arange(0, B).type_as(idx).view(-1, 1, 1) * N).view(-1)
    dims = x.size(1) // 3
    x = x.transpose(2, 1).contiguous()
    nbr = x.view(B * N, -1)[idx, :].view(B, N, k, dims, 3)
    ctr = x.view(B, N, 1, dims, 3).repeat(1, 1, k, 1, 1)
    cross = torch.cross(nbr, ctr, dim=-1)
    return torch.cat((nbr - ctr, ctr, cross), dim=3).permute(0, 3, 4, 1, 2).contiguous()


def _batchnorm(norm: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str, training: bool, momentum: float = 0.1):
    rm, rv = p[prefix + "running_mean"], p[prefix + "running_var"]
    return F.batch_norm(norm, rm, rv, p[prefix + "weight"], p[prefix + "bias"], training, momentum, 1e-5)


def vn_batchnorm(x: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str, dim: int, training: bool) -> torch.Tensor:
    """VNBatchNorm: x / ||x|| * BN(||x|| + EPS) over the 3-vector axis (dim 2).

    vector_neuron_layers.py:303-324; BatchNorm1d for dim 3/4 ("bn1d."), BatchNorm2d for dim 5 ("bn2d.").
    """
    norm = torch.norm(x, dim=2) + EPS
    sub = "bn2d." if dim == 5 else "bn1d."
    norm_bn = _batchnorm(norm, p, prefix + sub, training)
    return x / norm.unsqueeze(2) * norm_bn.unsqueeze(2)


def vn_linear_leaky_relu(x: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str, dim: int, training: bool,
                         negative_slope: float = 0.0) -> torch.Tensor:
    """VNLinearLeakyReLU: linear -> VN batch-norm -> direction-gated ReLU.

    vector_neuron_layers.py:251-273:  q = W_f x;  q = VNBN(q);  d = W_d x;
    out = slope*q + (1-slope) * (q if <q,d> >= 0 else q - <q,d>/(|d|^2+EPS) d).
    """
    q = F.linear(x.transpose(1, -1), p[prefix + "map_to_feat.weight"]).transpose(1, -1)
    q = vn_batchnorm(q, p, prefix + "batchnorm.", dim, training)
    d = F.linear(x.transpose(1, -1), p[prefix + "map_to_dir.weight"]).transpose(1, -1)
    dot = (q * d).sum(2, keepdim=True)
    mask = (dot >= 0).float()
    dsq = (d * d).sum(2, keepdim=True)
    return negative_slope * q + (1 - negative_slope) * (mask * q + (1 - mask) * (q - (dot / (dsq + EPS)) * d))


def vn_max_pool(x: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """VNMaxPool: pick, along the last axis, the sample maximising <x, W_d x> -- layers :349-364."""
    d = F.linear(x.transpose(1, -1), p[prefix + "map_to_dir.weight"]).transpose(1, -1)
    dot = (x * d).sum(2, keepdim=True)
    idx = dot.max(dim=-1, keepdim=False)[1]
    grids = torch.meshgrid([torch.arange(j) for j in x.size()[:-1]], indexing="ij") + (idx,)
    return x[grids]


def vnsmall_forward(point_cloud: torch.Tensor, p: Dict[str, torch.Tensor], n_knn: int = 20, pooling: str = "mean",
                    training: bool = False) -> torch.Tensor:
    """P2: VNSmall.forward, equivariant_networks.py:128-150.  (B, 3, N) -> (B, 3, 3).

    Dropout(0.5) is the identity in eval mode; the oracle supports ``training=True`` only for the
    batch-norm statistics (dropout stays off so the result is deterministic).
    """
    feat = graph_feature_cross(point_cloud.unsqueeze(1), k=n_knn)
    out = vn_linear_leaky_relu(feat, p, "conv_pos.", 5, training)
    if pooling == "mean":
        out = out.mean(dim=-1)
    elif pooling == "max":
        out = vn_max_pool(out, p, "pool.")
    else:
        raise ValueError(f"Pooling type {pooling} not supported")
    out = vn_linear_leaky_relu(out, p, "conv1.", 4, training)
    out = vn_batchnorm(out, p, "bn1.", 4, training)
    out = vn_linear_leaky_relu(out, p, "conv2.", 4, training)
    return out.mean(dim=-1)[:, :3]


def canonicalize_pointcloud(x: torch.Tensor, rotation: torch.Tensor) -> torch.Tensor:
    """P4: x_c = (x^T R^T)^T -- pointcloud/canonicalization/continuous_group.py:74-79."""
    return torch.bmm(x.transpose(1, 2), rotation.transpose(1, 2)).transpose(1, 2)


def continuous_prior_loss(rep: torch.Tensor) -> torch.Tensor:
    """P5: MSE(R, I) -- common/basecanonicalization.py:390-408."""
    eye = torch.eye(rep.shape[-1]).repeat(rep.shape[0], 1, 1)
    return torch.nn.MSELoss()(rep, eye)


def continuous_identity_metric(rep: torch.Tensor) -> torch.Tensor:
    """P5: 1 - MSE(R, I) -- common/basecanonicalization.py:410-430."""
    eye = torch.eye(rep.shape[-1]).repeat(rep.shape[0], 1, 1)
    return 1.0 - F.mse_loss(rep, eye).mean()


def nbody_invert(position: torch.Tensor, rotation: torch.Tensor, translation: torch.Tensor) -> torch.Tensor:
    """Row (f).4: x R + t, row-vector convention -- nbody/canonicalization/euclidean_group.py:126-137."""
    return torch.bmm(position[:, None, :], rotation).squeeze() + translation


def modified_gram_schmidt(vectors: torch.Tensor) -> torch.Tensor:
    """Row (f).4: nbody/canonicalization/euclidean_group.py:139-157 (second projection uses the updated vector)."""
    v1 = vectors[:, 0] / torch.norm(vectors[:, 0], dim=1, keepdim=True)
    v2 = vectors[:, 1] - torch.sum(vectors[:, 1] * v1, dim=1, keepdim=True) * v1
    v2 = v2 / torch.norm(v2, dim=1, keepdim=True)
    v3 = vectors[:, 2] - torch.sum(vectors[:, 2] * v1, dim=1, keepdim=True) * v1
    v3 = v3 - torch.sum(v3 * v2, dim=1, keepdim=True) * v2
    v3 = v3 / torch.norm(v3, dim=1, keepdim=True)
    return torch.stack([v1, v2, v3], dim=1)


def nbody_canonicalize(loc: torch.Tensor, vel: torch.Tensor, R: torch.Tensor, t: torch.Tensor):
    """Row (f).4: euclidean_group.py:108-124: loc R^-1 - t R^-1 and vel R^-1 with R^-1 = R^T (row vectors)."""
    Rinv = R.transpose(1, 2)
    cl = torch.bmm(loc[:, None, :], Rinv).squeeze() - torch.bmm(t[:, None, :], Rinv).squeeze()
    return cl, torch.bmm(vel[:, None, :], Rinv).squeeze()
