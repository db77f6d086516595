"""Vector-neuron layers used by VNSmall (reference: pointcloud/canonicalization_networks/vector_neuron_layers.py).

Only the layers on the hot path are provided (VNLinearLeakyReLU :210-273, VNBatchNorm :276-324,
VNMaxPool :327-364, mean_pool :367-380); module and parameter names match the reference state_dict.
Features are [B, C, 3, N, ...]: every channel is a 3-vector, linear maps mix channels only, so all layers
commute with rotations of the 3-axis.
"""
import torch
import torch.nn as nn

EPS = 1e-6


def _mix_channels(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """Apply a bias-free linear map over the channel axis (dim 1) of [B, C, 3, ...]."""
    W = lin.weight                                                   # (out, in)
    if x.is_cuda and W.shape[1] <= 4 and torch.is_grad_enabled():
        # conv_pos: 3 input channels on a (B, 3, 3, N, k) edge tensor.  As a GEMM this is 3.9 M rows x K = 3 (measured 5.8 ms
        # per map in the library at B = 64); as `in` fused multiply-adds over the output it is three streaming passes.
        # Training only: without autograd the GEMM form is kept, whose rounding the max-pooling golden vectors are pinned to
        # (an argmax over near-ties flips on a last-bit difference)
        shape = [1, W.shape[0]] + [1] * (x.dim() - 2)
        out = W[:, 0].view(shape) * x[:, 0:1]
        for i in range(1, W.shape[1]):
            out = torch.addcmul(out, W[:, i].view(shape), x[:, i:i + 1])
        return out
    return lin(x.transpose(1, -1)).transpose(1, -1)


class VNBatchNorm(nn.Module):
    """x / |x| * BN(|x| + EPS): batch-normalises the vector norms, keeps directions."""

    def __init__(self, num_features: int, dim: int):
        super().__init__()
        self.dim = dim
        if dim in (3, 4):
            self.bn1d = nn.BatchNorm1d(num_features)
        elif dim == 5:
            self.bn2d = nn.BatchNorm2d(num_features)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        norm = torch.norm(x, dim=2) + EPS
        norm_bn = self.bn2d(norm) if self.dim == 5 else self.bn1d(norm)
        return x / norm.unsqueeze(2) * norm_bn.unsqueeze(2)


class VNLinearLeakyReLU(nn.Module):
    """q = VNBN(W_f x); d = W_d x; keep q where <q,d> >= 0, else remove its component along d."""

    def __init__(self, in_channels: int, out_channels: int, dim: int = 5, share_nonlinearity: bool = False,
                 negative_slope: float = 0.2):
        super().__init__()
        self.dim = dim
        self.negative_slope = negative_slope
        self.map_to_feat = nn.Linear(in_channels, out_channels, bias=False)
        self.batchnorm = VNBatchNorm(out_channels, dim=dim)
        self.map_to_dir = nn.Linear(in_channels, 1 if share_nonlinearity else out_channels, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        q = self.batchnorm(_mix_channels(self.map_to_feat, x))
        d = _mix_channels(self.map_to_dir, x)
        dot = (q * d).sum(2, keepdim=True)
        keep = (dot >= 0).float()
        dsq = (d * d).sum(2, keepdim=True)
        s = self.negative_slope
        return s * q + (1 - s) * (keep * q + (1 - keep) * (q - (dot / (dsq + EPS)) * d))


class VNMaxPool(nn.Module):
    """Along the last axis pick the sample with the largest <x, W_d x> (per batch, channel, point)."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.map_to_dir = nn.Linear(in_channels, in_channels, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        d = _mix_channels(self.map_to_dir, x)
        idx = (x * d).sum(2, keepdim=True).max(dim=-1, keepdim=True)[1]
        return torch.gather(x, -1, idx.expand(*x.shape[:-1], 1)).squeeze(-1)


def mean_pool(x: torch.Tensor, dim: int = -1, keepdim: bool = False) -> torch.Tensor:
    return x.mean(dim=dim, keepdim=keepdim)
