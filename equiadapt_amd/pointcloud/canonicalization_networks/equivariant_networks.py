"""VNSmall: the SO(3)-equivariant canonicalization network for point clouds.

Reference: equiadapt/pointcloud/canonicalization_networks/equivariant_networks.py (knn :15-33,
get_graph_feature_cross :36-76, VNSmall :79-150).  Hyperparameters are read by attribute
(``n_knn``, ``pooling``), so a SimpleNamespace / dataclass / DictConfig all work.
"""
import os
from typing import Any, Optional

import torch
import torch.nn as nn

from equiadapt_amd.common.utils import mark_written, update_running_stats

from equiadapt_amd.pointcloud.canonicalization_networks.vector_neuron_layers import (
    VNBatchNorm,
    VNLinearLeakyReLU,
    VNMaxPool,
    mean_pool,
)


def knn(x: torch.Tensor, k: int) -> torch.Tensor:
    """(B, 3, N) -> (B, N, k) indices of the k nearest points (self included), by -||xi-xj||^2 top-k."""
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x**2, dim=1, keepdim=True)
    return (-xx - inner - xx.transpose(2, 1)).topk(k=k, dim=-1)[1]


def get_graph_feature_cross(x: torch.Tensor, k: int = 20, idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B, 1, 3, N) -> (B, 3, 3, N, k) edge features [neighbour - centre, centre, neighbour x centre]."""
    B, N = x.size(0), x.size(3)
    x = x.view(B, -1, N)
    if idx is None:
        idx = knn(x, k=k)
    flat = (idx + torch.arange(0, B, device=idx.device).view(-1, 1, 1) * N).view(-1)
    dims = x.size(1) // 3
    pts = x.transpose(2, 1).contiguous()
    nbr = pts.view(B * N, -1)[flat, :].view(B, N, k, dims, 3)
    ctr = pts.view(B, N, 1, dims, 3).expand(B, N, k, dims, 3)
    cross = torch.cross(nbr, ctr, dim=-1)
    return torch.cat((nbr - ctr, ctr, cross), dim=3).permute(0, 3, 4, 1, 2).contiguous()


class ConvPosMeanPool(torch.autograd.Function):
    """mean over the k neighbours of VNLinearLeakyReLU(3 -> 21, slope 0)(edge features), with autograd w.r.t. the layer's
    parameters, on the recompute-everything kernels of csrc/vnsmall_train.hip (eqa_vn_*): (B, 3, N) -> (B, 21, 3, N).
    Training-mode VN batch-norm: statistics of n = |W_f f| + EPS over all B*N*k edges (fp64 accumulation across blocks),
    running statistics updated like nn.BatchNorm2d.  No gradient w.r.t. the cloud."""

    @staticmethod
    def forward(ctx, x, Wf, Wd, gamma, beta, bn, k):
        from equiadapt_amd import _lib, ops

        lib = _lib.load()
        B, _, N = x.shape
        x = x.contiguous()
        Wf_, Wd_ = Wf.detach().contiguous(), Wd.detach().contiguous()
        C = Wf_.shape[0]
        st = ops._stream()
        nblk = B * lib.eqa_vn_blocks(N)
        dev = x.device
        with torch.cuda.device(dev):
            idx = torch.empty((B, N, k), dtype=torch.int32, device=dev)
            _lib.check(lib.eqa_vn_knn(x.data_ptr(), idx.data_ptr(), B, N, k, st), "eqa_vn_knn")
            M = B * N * k
            batch_stats = bool(bn.training or bn.running_mean is None)
            stat = torch.empty(128, dtype=torch.float32, device=dev)          # scale | shift | mean | rstd, 32 floats each
            if batch_stats and (bn.momentum is not None or not bn.track_running_stats):
                part = torch.empty((nblk, C, 2), dtype=torch.float32, device=dev)
                _lib.check(lib.eqa_vn_convpos_stats(x.data_ptr(), idx.data_ptr(), Wf_.data_ptr(), part.data_ptr(), B, N, k, st),
                           "eqa_vn_convpos_stats")
                track = bn.track_running_stats and bn.running_mean is not None
                _lib.check(lib.eqa_vn_bn_finalize(part.data_ptr(), nblk, 2 * C, C, M, gamma.detach().data_ptr(), beta.detach().data_ptr(),
                                                  bn.running_mean.data_ptr() if track else None,
                                                  bn.running_var.data_ptr() if track else None,
                                                  bn.num_batches_tracked.data_ptr() if track else None,
                                                  float(bn.momentum or 0.0), float(bn.eps), stat.data_ptr(), st), "eqa_vn_bn_finalize")
                if track:
                    mark_written(bn.running_mean, bn.running_var, bn.num_batches_tracked)
            else:
                if batch_stats:                                                # cumulative moving average (momentum=None): host glue
                    part = torch.empty((nblk, C, 2), dtype=torch.float32, device=dev)
                    _lib.check(lib.eqa_vn_convpos_stats(x.data_ptr(), idx.data_ptr(), Wf_.data_ptr(), part.data_ptr(), B, N, k, st),
                               "eqa_vn_convpos_stats")
                    sums = part.sum(0, dtype=torch.float64)
                    mean = sums[:, 0] / M
                    var = (sums[:, 1] / M - mean * mean).clamp_min(0.0)
                    update_running_stats(bn, mean, var * (M / max(M - 1, 1)))
                    mean, var = mean.float(), var.float()
                else:
                    mean, var = bn.running_mean, bn.running_var
                rstd = torch.rsqrt(var + bn.eps)
                sc = gamma.detach() * rstd
                stat[0:C], stat[32:32 + C], stat[64:64 + C], stat[96:96 + C] = sc, beta.detach() - mean * sc, mean, rstd
            scale, shift, mean, rstd = stat[0:C], stat[32:32 + C], stat[64:64 + C], stat[96:96 + C]
            pooled = torch.empty((B, C, 3, N), dtype=torch.float32, device=dev)
            _lib.check(lib.eqa_vn_convpos_fwd(x.data_ptr(), idx.data_ptr(), Wf_.data_ptr(), Wd_.data_ptr(), scale.data_ptr(),
                                              shift.data_ptr(), pooled.data_ptr(), B, N, k, st), "eqa_vn_convpos_fwd")
        ctx.save_for_backward(x, idx, Wf_, Wd_, stat)
        ctx.batch_stats, ctx.M, ctx.k = batch_stats, M, k
        return pooled

    @staticmethod
    def backward(ctx, gpool):
        from equiadapt_amd import _lib, ops

        lib = _lib.load()
        x, idx, Wf, Wd, stat = ctx.saved_tensors
        B, _, N = x.shape
        C, k = Wf.shape[0], ctx.k
        scale, shift, mean, rstd = stat[0:C], stat[32:32 + C], stat[64:64 + C], stat[96:96 + C]
        gpool = gpool.contiguous()
        st = ops._stream()
        nblk = B * lib.eqa_vn_blocks(N)
        dev = x.device
        with torch.cuda.device(dev):
            part = torch.empty((nblk, C, 2), dtype=torch.float32, device=dev)
            _lib.check(lib.eqa_vn_convpos_bwd_reduce(x.data_ptr(), idx.data_ptr(), Wf.data_ptr(), Wd.data_ptr(), scale.data_ptr(),
                                                     shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gpool.data_ptr(),
                                                     part.data_ptr(), B, N, k, st), "eqa_vn_convpos_bwd_reduce")
            grads = torch.empty(64, dtype=torch.float32, device=dev)            # d beta[32] | d gamma[32]
            red = torch.empty(64, dtype=torch.float32, device=dev)              # m1[32] | m2[32]
            _lib.check(lib.eqa_vn_bn_bwd_finalize(part.data_ptr(), nblk, 2 * C, C, ctx.M, grads.data_ptr(), red.data_ptr(), st),
                       "eqa_vn_bn_bwd_finalize")
            dbeta, dgamma = grads[:C], grads[32:32 + C]
            if not ctx.batch_stats:
                red.zero_()
            wpart = torch.empty((nblk, C, 6), dtype=torch.float32, device=dev)
            _lib.check(lib.eqa_vn_convpos_bwd_apply(x.data_ptr(), idx.data_ptr(), Wf.data_ptr(), Wd.data_ptr(), scale.data_ptr(),
                                                    shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), red.data_ptr(),
                                                    red[32:].data_ptr(), gpool.data_ptr(), wpart.data_ptr(), B, N, k, st),
                       "eqa_vn_convpos_bwd_apply")
            dW = wpart.sum(0, dtype=torch.float64).float()
        return None, dW[:, :3].contiguous(), dW[:, 3:].contiguous(), dgamma, dbeta, None, None


class TailMean(torch.autograd.Function):
    """mean over the points of dropout(conv2(bn1(conv1(pooled)))) with the three VN batch-norms in training mode, on the
    recompute-everything passes of csrc/vnsmall_tail.hip (eqa_vn_tail_pass): (B, 21, 3, N) -> (B, 4, 3), with autograd
    w.r.t. the pooled features and the ten parameters.  Statistics of the vector norms over all B*N points (fp64 across
    blocks); running statistics and num_batches_tracked updated like nn.BatchNorm1d.  ``mask``: the dropout factors
    (0 or 1/(1-p)) of the (B, 4, 3, N) output of conv2, or None."""

    @staticmethod
    def forward(ctx, pooled, Wf1, Wd1, Wf2, Wd2, g1, b1, g2, b2, g3, b3, bns, mask):
        from equiadapt_amd import _lib, ops

        lib = _lib.load()
        B, C, _, N = pooled.shape
        dev = pooled.device
        pooled = pooled.detach().contiguous()
        W = torch.cat([w.detach().reshape(-1) for w in (Wf1, Wd1, Wf2, Wd2)]).contiguous()
        assert C == 21 and W.numel() == 1050
        nbx = lib.eqa_vn_tail_blocks(N)
        nblk = B * nbx
        st = ops._stream()
        p_mask = mask.data_ptr() if mask is not None else None
        with torch.cuda.device(dev):
            stat = torch.zeros((3, 128), dtype=torch.float32, device=dev)
            part = torch.empty(nblk * 42, dtype=torch.float32, device=dev)
            for layer, (bn, gamma, beta) in enumerate(zip(bns, (g1, g2, g3), (b1, b2, b3))):
                _lib.check(lib.eqa_vn_tail_pass(layer, pooled.data_ptr(), W.data_ptr(), stat.data_ptr(), None, None, None,
                                                part.data_ptr(), None, B, N, st), "eqa_vn_tail_pass")
                track = bn.track_running_stats and bn.running_mean is not None
                _lib.check(lib.eqa_vn_bn_finalize(part.data_ptr(), nblk, lib.eqa_vn_tail_partial_floats(layer), gamma.numel(), B * N,
                                                  gamma.detach().data_ptr(), beta.detach().data_ptr(),
                                                  bn.running_mean.data_ptr() if track else None,
                                                  bn.running_var.data_ptr() if track else None,
                                                  bn.num_batches_tracked.data_ptr() if track else None,
                                                  float(bn.momentum), float(bn.eps), stat[layer].data_ptr(), st), "eqa_vn_bn_finalize")
                if track:
                    mark_written(bn.running_mean, bn.running_var, bn.num_batches_tracked)
            _lib.check(lib.eqa_vn_tail_pass(3, pooled.data_ptr(), W.data_ptr(), stat.data_ptr(), None, p_mask, None,
                                            part.data_ptr(), None, B, N, st), "eqa_vn_tail_pass")
            out = part[:nblk * 12].view(B, nbx, 12).sum(1).div_(N).view(B, 4, 3)
        ctx.save_for_backward(pooled, W, stat, mask)
        return out

    @staticmethod
    def backward(ctx, gout):
        from equiadapt_amd import _lib, ops

        lib = _lib.load()
        pooled, W, stat, mask = ctx.saved_tensors
        B, _, _, N = pooled.shape
        dev = pooled.device
        gout = gout.contiguous().float()
        nblk = B * lib.eqa_vn_tail_blocks(N)
        st = ops._stream()
        p_mask = mask.data_ptr() if mask is not None else None
        with torch.cuda.device(dev):
            red = torch.zeros((3, 64), dtype=torch.float32, device=dev)
            grads = torch.empty((3, 64), dtype=torch.float32, device=dev)
            part = torch.empty(nblk * 483, dtype=torch.float32, device=dev)
            g_pooled = torch.empty_like(pooled)
            dW2 = dWd1 = None
            for pas, layer, C in ((4, 2, 4), (5, 1, 21), (6, 0, 21)):
                _lib.check(lib.eqa_vn_tail_pass(pas, pooled.data_ptr(), W.data_ptr(), stat.data_ptr(), red.data_ptr(), p_mask,
                                                gout.data_ptr(), part.data_ptr(), None, B, N, st), "eqa_vn_tail_pass")
                stride = lib.eqa_vn_tail_partial_floats(pas)
                _lib.check(lib.eqa_vn_bn_bwd_finalize(part.data_ptr(), nblk, stride, C, B * N, grads[layer].data_ptr(),
                                                      red[layer].data_ptr(), st), "eqa_vn_bn_bwd_finalize")
                if pas != 4:   # weight-gradient partials behind the 42 sums: conv2 (168) in pass 5, conv1.map_to_dir (441) in pass 6
                    dW = part[:nblk * stride].view(nblk, stride)[:, 42:].sum(0, dtype=torch.float64).float()
                    dW2, dWd1 = (dW, dWd1) if pas == 5 else (dW2, dW)
            _lib.check(lib.eqa_vn_tail_pass(7, pooled.data_ptr(), W.data_ptr(), stat.data_ptr(), red.data_ptr(), p_mask,
                                            gout.data_ptr(), part.data_ptr(), g_pooled.data_ptr(), B, N, st), "eqa_vn_tail_pass")
            dWf1 = part[:nblk * 441].view(nblk, 441).sum(0, dtype=torch.float64).float()
        db = [grads[i, :c] for i, c in ((0, 21), (1, 21), (2, 4))]
        dg = [grads[i, 32:32 + c] for i, c in ((0, 21), (1, 21), (2, 4))]
        return (g_pooled, dWf1.view(21, 21), dWd1.view(21, 21), dW2[:84].view(4, 21), dW2[84:].view(4, 21),
                dg[0], db[0], dg[1], db[1], dg[2], db[2], None, None)


class VNSmall(nn.Module):
    """(B, 3, N) point cloud -> (B, 3, 3): three vectors that rotate with the cloud."""

    def __init__(self, hyperparams: Any):
        super().__init__()
        self.n_knn = hyperparams.n_knn
        self.pooling = hyperparams.pooling
        self.conv_pos = VNLinearLeakyReLU(3, 64 // 3, dim=5, negative_slope=0.0)
        self.conv1 = VNLinearLeakyReLU(64 // 3, 64 // 3, dim=4, negative_slope=0.0)
        self.bn1 = VNBatchNorm(64 // 3, dim=4)
        self.conv2 = VNLinearLeakyReLU(64 // 3, 12 // 3, dim=4, negative_slope=0.0)
        self.dropout = nn.Dropout(p=0.5)
        if self.pooling == "max":
            self.pool = VNMaxPool(64 // 3)
        elif self.pooling == "mean":
            self.pool = mean_pool  # type: ignore
        else:
            raise ValueError(f"Pooling type {self.pooling} not supported")

    def packed_parameters(self) -> torch.Tensor:
        """The 1310 floats the fused kernel consumes (layout: csrc/pointcloud.hip), eval-mode batch-norms folded to a
        scale/shift of the vector norm; "max" pooling: + the 441 of the pooling layer's direction map.  Cached per parameter
        version."""
        tensors = list(self.parameters()) + [b for b in self.buffers()]
        key = tuple(t._version for t in tensors) + (str(tensors[0].device),)
        hit = getattr(self, "_packed", None)
        if hit is not None and hit[0] == key:
            return hit[1]

        def fold(bn):
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            return scale, bn.bias - bn.running_mean * scale

        parts = []
        for lay, bn in ((self.conv_pos, self.conv_pos.batchnorm.bn2d), (self.conv1, self.conv1.batchnorm.bn1d)):
            sc, sh = fold(bn)
            parts += [lay.map_to_feat.weight.flatten(), lay.map_to_dir.weight.flatten(), sc, sh]
        parts += list(fold(self.bn1.bn1d))
        sc, sh = fold(self.conv2.batchnorm.bn1d)
        parts += [self.conv2.map_to_feat.weight.flatten(), self.conv2.map_to_dir.weight.flatten(), sc, sh]
        if self.pooling == "max":
            parts.append(self.pool.map_to_dir.weight.flatten())
        packed = torch.cat([p.detach().float() for p in parts]).contiguous()
        assert packed.numel() == (1751 if self.pooling == "max" else 1310)
        self._packed = (key, packed)
        return packed

    def fused_inference_applies(self, point_cloud: torch.Tensor) -> bool:
        """eval mode without autograd on the device, at a size the fused kernel takes (eqa_vnsmall_fwd / _canonicalize)."""
        return bool(point_cloud.is_cuda and not self.training and not torch.is_grad_enabled() and point_cloud.dim() == 3
                    and 1 <= self.n_knn <= 32 and self.n_knn <= point_cloud.shape[-1] <= 6144 and point_cloud.dtype == torch.float32)

    def forward(self, point_cloud: torch.Tensor) -> torch.Tensor:
        if self.fused_inference_applies(point_cloud):
            from equiadapt_amd import ops

            return ops.vnsmall_forward(point_cloud, self.packed_parameters(), self.n_knn, self.pooling)
        if (point_cloud.is_cuda and torch.is_grad_enabled() and not point_cloud.requires_grad and self.pooling == "mean"
                and 1 <= self.n_knn <= 32 and self.n_knn <= point_cloud.shape[-1] <= 4096 and point_cloud.dtype == torch.float32
                and os.environ.get("EQA_TRAIN_FAST", "1") != "0"):
            # training: the (B, 21, 3, N, k) edge tensors of the first block are never materialised (csrc/vnsmall_train.hip)
            cp = self.conv_pos
            out = ConvPosMeanPool.apply(point_cloud, cp.map_to_feat.weight, cp.map_to_dir.weight, cp.batchnorm.bn2d.weight,
                                        cp.batchnorm.bn2d.bias, cp.batchnorm.bn2d, self.n_knn)
        else:
            feat = get_graph_feature_cross(point_cloud.unsqueeze(1), k=self.n_knn)
            out = self.pool(self.conv_pos(feat))
        bns = (self.conv1.batchnorm.bn1d, self.bn1.bn1d, self.conv2.batchnorm.bn1d)
        if (out.is_cuda and self.training and torch.is_grad_enabled() and out.dtype == torch.float32 and out.shape[1] == 21
                and out.shape[0] <= 65535 and all(bn.training and bn.momentum is not None and bn.affine for bn in bns)
                and os.environ.get("EQA_TRAIN_FAST", "1") != "0"):
            # training: conv1 -> bn1 -> conv2 -> dropout -> mean in eight recompute passes (csrc/vnsmall_tail.hip)
            mask = None
            if self.dropout.p > 0:
                B, _, _, N = out.shape
                mask = nn.functional.dropout(torch.ones((B, 4, 3, N), dtype=out.dtype, device=out.device), self.dropout.p, True)
            y = TailMean.apply(out, self.conv1.map_to_feat.weight, self.conv1.map_to_dir.weight, self.conv2.map_to_feat.weight,
                               self.conv2.map_to_dir.weight, bns[0].weight, bns[0].bias, bns[1].weight, bns[1].bias,
                               bns[2].weight, bns[2].bias, bns, mask)
            return y[:, :3]
        out = self.bn1(self.conv1(out))
        out = self.dropout(self.conv2(out))
        return out.mean(dim=-1)[:, :3]
