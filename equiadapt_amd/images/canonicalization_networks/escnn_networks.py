"""ESCNNEquivariantNetwork: regular-representation G-CNN canonicalization network.

Reference: equiadapt/images/canonicalization_networks/escnn_networks.py:8-117, built there from e2cnn
``R2Conv`` + ``InnerBatchNorm(momentum=0.9)`` + ``ReLU`` + ``PointwiseDropout(0.5)``.

e2cnn is not available to this build and its steerable-basis expansion is not restated here
(SURVEY.md section 8c).  This class keeps the reference's constructor, tensor shapes, layer order, FLOPs and
output contract ((B, G) activations = mean over fields and space), and parameterises every conv by a
rotated filter bank (trivial->regular lifting, then regular->regular group convs, all k x k, no padding).
InnerBatchNorm over a regular field shares statistics across the G channels of the field, i.e. it is
BatchNorm3d over (B, fields, G, H, W); PointwiseDropout is element-wise dropout.
Weights trained with e2cnn can be brought in through their exported dense form
(``R2Conv.export()`` -> Conv2d, ``InnerBatchNorm.export()`` -> BatchNorm2d): ``load_exported_dense``.
"""
import os
from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from equiadapt_amd.images.canonicalization_networks.custom_group_equivariant_layers import (
    RotationEquivariantConv,
    RotationEquivariantConvLift,
    RotoReflectionEquivariantConv,
    RotoReflectionEquivariantConvLift,
)
from equiadapt_amd import ops
from equiadapt_amd.common.utils import update_running_stats
from equiadapt_amd.images.canonicalization_networks import fftconv, winograd
from equiadapt_amd.images.canonicalization_networks.pooling import (
    WindowSumsFunction,
    conv_then_group_pool,
    group_pool,
    window_sums_to_activations,
)


class _InnerBatchNorm(nn.BatchNorm3d):
    """Per-field batch norm over (batch, group, space) of a (B, fields, G, H, W) map."""


class LiftConvFunction(torch.autograd.Function):
    """Training counterpart of the lifting convolution in `_forward_inference`: forward on the fp32-MFMA kernel
    (`eqa_lift_conv_nhwc`, 0.71 ms at the headline shape; wide / single-channel filters: `eqa_lift_conv_wide`), filter gradient on
    the matrix cores too (`eqa_lift_conv_wgrad_nhwc` / `eqa_lift_conv_wide_wgrad`; shapes neither takes: the framework's
    convolution-weight-gradient).  The input image needs no gradient in the canonicalizer; if asked for, it is the framework's."""

    @staticmethod
    def forward(ctx, x, bank, with_stats=False):
        """with_stats: also return the fp64 partial sums (rows, Cout, 2) of the output's per-channel sum / sum of squares, taken in the
        kernel's epilogue (eqa_lift_conv_nhwc_stats) -- the batch-norm behind the layer then skips its own pass over the map."""
        ctx.save_for_backward(x, bank)
        k = bank.shape[-1]
        if not ops.lift_conv_supported(bank.shape[1], bank.shape[-2], k, bank.shape[0]):      # wide / single-channel filters
            if with_stats:
                raise ValueError("LiftConvFunction: with_stats needs a shape eqa_lift_conv_nhwc_stats takes (gate on ops.lift_conv_supported)")
            return ops.lift_conv_wide(x, ops.pack_lift_weights_wide(bank.detach()), None, False, bank.shape[-2], k)
        wpk = ops.pack_lift_weights(bank.detach())
        if with_stats:
            y, part = ops.lift_conv_nhwc_stats(x, wpk, bank.shape[-2], k)
            ctx.mark_non_differentiable(part)
            return y, part
        return ops.lift_conv_nhwc(x, wpk, None, False, bank.shape[-2], k)

    @staticmethod
    def backward(ctx, dy, *_):
        x, bank = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.nn.grad.conv2d_input(x.shape, bank, dy) if ctx.needs_input_grad[0] else None
        dbank = None
        if ctx.needs_input_grad[1]:
            kh, kw = bank.shape[-2], bank.shape[-1]
            if ops.lift_conv_wgrad_supported(x, bank.shape[0], kh, kw):
                dbank = ops.lift_conv_wgrad_nhwc(x, dy, kh, kw)            # fp32 MFMA, 0.8 ms where MIOpen's solvers take 3.1
            elif ops.lift_conv_wide_supported(bank.shape[1], kh, kw, bank.shape[0]) and x.shape[0] > 0:
                dbank = ops.lift_conv_wide_wgrad(x, dy, kh, kw)            # the wide / single-channel filters (tutorial k = 9)
            else:
                dbank = torch.nn.grad.conv2d_weight(x, bank.shape, dy)
        return dx, dbank, None


class InnerBnReluDropout(torch.autograd.Function):
    """dropout(relu(InnerBatchNorm(h))) on a channels-last (B, fields*E, H, W) map, forward and backward on the fused
    kernels of csrc/batchnorm.hip (eqa_bn_*).  Statistics per field over (batch, group, space) in fp64; running statistics
    updated like nn.BatchNorm3d (momentum, unbiased variance).  The preceding convolution's bias only shifts the mean (it
    cancels in the normalised output), so it enters the running mean and nowhere else.  The dropout mask is a hash of a
    seed drawn from torch's CPU generator; the backward recomputes "kept and positive" from h, the affine map and that seed
    instead of reading y (one map less per pass)."""

    @staticmethod
    def forward(ctx, h, weight, bias, bn, E, conv_bias, p, drop_training, part=None):
        """part: (rows, C, 2) fp64 partial sums of h's per-channel sum / sum of squares when the kernel that produced h took them
        on the way out (LiftConvFunction with_stats, the FFT convolution's inverse transform); None: one pass over h here."""
        from equiadapt_amd import _lib, ops

        lib = _lib.load()
        Bn, C, H, W = h.shape
        Fd = C // E
        npix = Bn * H * W
        st = ops._stream()
        with torch.cuda.device(h.device):
            if bn.training or bn.running_mean is None:
                if part is None:
                    nblk = lib.eqa_bn_partial_blocks(npix)
                    part = torch.empty((nblk, C, 2), dtype=torch.float64, device=h.device)
                    _lib.check(lib.eqa_bn_stats_nhwc(h.data_ptr(), part.data_ptr(), npix, C, st), "eqa_bn_stats_nhwc")
                sums = part.sum(0).view(Fd, E, 2).sum(1)                         # (fields, 2) fp64
                n = npix * E
                mean = sums[:, 0] / n
                var = (sums[:, 1] / n - mean * mean).clamp_min(0.0)
                full_mean = mean if conv_bias is None else mean + conv_bias.detach().double()
                update_running_stats(bn, full_mean, var * (n / max(n - 1, 1)))
                mean, var = mean.float(), var.float()
            else:
                mean = bn.running_mean if conv_bias is None else bn.running_mean - conv_bias.detach()
                var = bn.running_var
            rstd = torch.rsqrt(var + bn.eps)
            scale_f = weight.detach() * rstd
            scale = scale_f.repeat_interleave(E).contiguous()
            shift = (bias.detach() - mean * scale_f).repeat_interleave(E).contiguous()
            p_eff = float(p) if drop_training else 0.0
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_eff > 0 else 0
            y = torch.empty_like(h)
            _lib.check(lib.eqa_bn_relu_dropout_nhwc(h.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), npix, C, p_eff,
                                                    seed, st), "eqa_bn_relu_dropout_nhwc")
        # y is not kept for the backward: "kept and positive" is recomputed from h, scale, shift and the seed
        ctx.save_for_backward(h, weight, mean.repeat_interleave(E).contiguous(), rstd.repeat_interleave(E).contiguous(), scale, shift)
        ctx.E, ctx.p, ctx.seed, ctx.batch_stats = E, p_eff, seed, bool(bn.training or bn.running_mean is None)
        ctx.has_conv_bias = conv_bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        from equiadapt_amd import _lib, ops

        lib = _lib.load()
        h, weight, mean_c, rstd_c, scale, shift = ctx.saved_tensors
        E = ctx.E
        Bn, C, H, W = h.shape
        Fd = C // E
        npix = Bn * H * W
        gy = gy.contiguous(memory_format=torch.channels_last)
        st = ops._stream()
        with torch.cuda.device(h.device):
            nblk = lib.eqa_bn_partial_blocks(npix)
            part = torch.empty((nblk, C, 2), dtype=torch.float64, device=h.device)
            _lib.check(lib.eqa_bn_bwd_reduce_nhwc(gy.data_ptr(), None, h.data_ptr(), mean_c.data_ptr(), rstd_c.data_ptr(), ctx.p,
                                                  part.data_ptr(), npix, C, scale.data_ptr(), shift.data_ptr(), ctx.seed, st),
                       "eqa_bn_bwd_reduce_nhwc")
            sums = part.sum(0).view(Fd, E, 2).sum(1)                             # per field: sum g, sum g*xhat
            dbias, dweight = sums[:, 0].float(), sums[:, 1].float()
            n = npix * E
            a = (weight * rstd_c.view(Fd, E)[:, 0]).repeat_interleave(E).contiguous()
            if ctx.batch_stats:
                b = (sums[:, 0] / n).float().repeat_interleave(E).contiguous()
                d = (sums[:, 1] / n).float().repeat_interleave(E).contiguous()
            else:                                                                # running statistics are constants
                b = torch.zeros(C, device=h.device)
                d = b
            dh = torch.empty_like(h)
            _lib.check(lib.eqa_bn_bwd_apply_nhwc(gy.data_ptr(), None, h.data_ptr(), mean_c.data_ptr(), rstd_c.data_ptr(),
                                                 a.data_ptr(), b.data_ptr(), d.data_ptr(), ctx.p, dh.data_ptr(), npix, C,
                                                 scale.data_ptr(), shift.data_ptr(), ctx.seed, st), "eqa_bn_bwd_apply_nhwc")
        # With batch statistics the convolution bias cancels in the normalised output (zero gradient).  With running statistics
        # (frozen batch-norm under autograd) it does not: out = scale * (h + conv_bias - running_mean) + ..., so
        # d conv_bias = per-field sum of dh, as the op-by-op path and the reference propagate it.
        # The zero is returned as a tensor, not as None: the reference and the op-by-op path leave an exact-zero .grad on the
        # bias, which weight decay then acts on and which DistributedDataParallel needs to count the parameter as used.
        dconv_bias = None
        if ctx.has_conv_bias and ctx.needs_input_grad[5]:
            if ctx.batch_stats:
                dconv_bias = torch.zeros(Fd, dtype=h.dtype, device=h.device)
            else:
                dconv_bias = dh.sum(dim=(0, 2, 3), dtype=torch.float64).view(Fd, E).sum(1).float()
        return dh, dweight, dbias, None, None, dconv_bias, None, None, None


def _window_grad_table(dS: torch.Tensor, H: int, W: int, k: int) -> torch.Tensor:
    """Backward of the k*k shifted-window sums as a table: a pixel's gradient depends only on the class of its row and of its
    column (the k-1 top / left border indices, the interior, the k-1 bottom / right ones): (B, C, k, k) -> (B, 2k-1, 2k-1, C) fp32."""
    nb, dev = k - 1, dS.device
    if (dS.is_cuda and dS.dtype == torch.float64 and k <= ops.MAX_WINDOW_K and dS.shape[0] <= 65535 and H >= 2 * nb + 1 and W >= 2 * nb + 1
            and os.environ.get("EQA_WS_TABLE_KERNEL", "1") != "0"):
        return ops.window_grad_table(dS.contiguous(), H, W)       # rectangle sums from prefix sums: one launch, no library GEMM

    def mask(n):
        rep = torch.cat([torch.arange(nb), torch.tensor([nb]), torch.arange(n - nb, n)]).to(dev)       # one representative per class
        u = torch.arange(k, device=dev)
        return ((u[None, :] <= rep[:, None]) & (rep[:, None] <= u[None, :] + (n - k))).to(dS.dtype)    # (2nb+1, k)

    return torch.einsum("tu,bcuv,sv->btsc", mask(H), dS, mask(W)).float().contiguous()


class InnerBnReluDropoutWindowSums(torch.autograd.Function):
    """The last hidden block and the window sums that consume it, S = window_sums(dropout(relu(InnerBatchNorm(h))), k), without
    the block's output ever existing: the forward applies the affine map, the ReLU and the dropout mask while the window-sum
    kernel loads h (eqa_window_sums_nhwc_act); the backward reads the upstream gradient from the window sums' (2k-1) x (2k-1) class
    table (eqa_bn_bwd_*_wsgrad) instead of an expanded map.  Against InnerBnReluDropout + WindowSumsFunction at the headline shape
    (256 x 88 x 88 x 256): one 2.2 GB map less written and read in the forward, one less written and two fewer reads in the backward
    (-2.0 ms of the 29.7 ms training step).  Same statistics, same mask, same arithmetic per element."""

    @staticmethod
    def forward(ctx, h, weight, bias, bn, E, conv_bias, p, drop_training, k, part=None):
        from equiadapt_amd import _lib, ops

        lib = _lib.load()
        Bn, C, H, W = h.shape
        Fd = C // E
        npix = Bn * H * W
        st = ops._stream()
        with torch.cuda.device(h.device):
            batch_stats = bool(bn.training or bn.running_mean is None)
            if batch_stats:
                if part is None:                                                 # (else: taken by the producer of h, see InnerBnReluDropout)
                    nblk = lib.eqa_bn_partial_blocks(npix)
                    part = torch.empty((nblk, C, 2), dtype=torch.float64, device=h.device)
                    _lib.check(lib.eqa_bn_stats_nhwc(h.data_ptr(), part.data_ptr(), npix, C, st), "eqa_bn_stats_nhwc")
                sums = part.sum(0).view(Fd, E, 2).sum(1)
                n = npix * E
                mean = sums[:, 0] / n
                var = (sums[:, 1] / n - mean * mean).clamp_min(0.0)
                full_mean = mean if conv_bias is None else mean + conv_bias.detach().double()
                update_running_stats(bn, full_mean, var * (n / max(n - 1, 1)))
                mean, var = mean.float(), var.float()
            else:
                mean = bn.running_mean if conv_bias is None else bn.running_mean - conv_bias.detach()
                var = bn.running_var
            rstd = torch.rsqrt(var + bn.eps)
            scale_f = weight.detach() * rstd
            scale = scale_f.repeat_interleave(E).contiguous()
            shift = (bias.detach() - mean * scale_f).repeat_interleave(E).contiguous()
            p_eff = float(p) if drop_training else 0.0
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_eff > 0 else 0
            S = torch.empty((Bn, C, k, k), dtype=torch.float64, device=h.device)
            ws = torch.empty((max(lib.eqa_window_sums_nhwc_workspace_bytes(Bn, C, H, k), 4) // 4,), dtype=torch.float32, device=h.device)
            _lib.check(lib.eqa_window_sums_nhwc_act(h.data_ptr(), scale.data_ptr(), shift.data_ptr(), 1, p_eff, seed, S.data_ptr(),
                                                    ws.data_ptr(), Bn, C, H, W, k, st), "eqa_window_sums_nhwc_act")
        ctx.save_for_backward(h, weight, mean.repeat_interleave(E).contiguous(), rstd.repeat_interleave(E).contiguous(), scale, shift)
        ctx.E, ctx.p, ctx.seed, ctx.batch_stats, ctx.k = E, p_eff, seed, batch_stats, k
        ctx.has_conv_bias = conv_bias is not None
        return S

    @staticmethod
    def backward(ctx, dS):
        from equiadapt_amd import _lib, ops

        lib = _lib.load()
        h, weight, mean_c, rstd_c, scale, shift = ctx.saved_tensors
        E, k = ctx.E, ctx.k
        Bn, C, H, W = h.shape
        Fd = C // E
        npix = Bn * H * W
        table = _window_grad_table(dS, H, W, k)
        st = ops._stream()
        with torch.cuda.device(h.device):
            nblk = lib.eqa_bn_partial_blocks(npix)
            part = torch.empty((nblk, C, 2), dtype=torch.float64, device=h.device)
            _lib.check(lib.eqa_bn_bwd_reduce_nhwc_wsgrad(table.data_ptr(), h.data_ptr(), mean_c.data_ptr(), rstd_c.data_ptr(), ctx.p,
                                                         part.data_ptr(), Bn, H, W, C, k, scale.data_ptr(), shift.data_ptr(), ctx.seed, st),
                       "eqa_bn_bwd_reduce_nhwc_wsgrad")
            sums = part.sum(0).view(Fd, E, 2).sum(1)
            dbias, dweight = sums[:, 0].float(), sums[:, 1].float()
            n = npix * E
            a = (weight * rstd_c.view(Fd, E)[:, 0]).repeat_interleave(E).contiguous()
            if ctx.batch_stats:
                b = (sums[:, 0] / n).float().repeat_interleave(E).contiguous()
                d = (sums[:, 1] / n).float().repeat_interleave(E).contiguous()
            else:
                b = torch.zeros(C, device=h.device)
                d = b
            dh = torch.empty_like(h)
            _lib.check(lib.eqa_bn_bwd_apply_nhwc_wsgrad(table.data_ptr(), h.data_ptr(), mean_c.data_ptr(), rstd_c.data_ptr(), a.data_ptr(),
                                                        b.data_ptr(), d.data_ptr(), ctx.p, dh.data_ptr(), Bn, H, W, C, k, scale.data_ptr(),
                                                        shift.data_ptr(), ctx.seed, st), "eqa_bn_bwd_apply_nhwc_wsgrad")
        dconv_bias = None
        if ctx.has_conv_bias and ctx.needs_input_grad[5]:
            if ctx.batch_stats:
                dconv_bias = torch.zeros(Fd, dtype=h.dtype, device=h.device)
            else:
                dconv_bias = dh.sum(dim=(0, 2, 3), dtype=torch.float64).view(Fd, E).sum(1).float()
        return dh, dweight, dbias, None, None, dconv_bias, None, None, None, None


class _DenseConv(nn.Module):
    """An exported dense convolution (e2cnn ``R2Conv.export()`` -> nn.Conv2d over fields x group channels, channel index =
    field * |G| + element) behind the interface the inference fast path expects from a group convolution."""

    def __init__(self, conv: nn.Conv2d, num_group_elements: int, lifting: bool):
        super().__init__()
        if conv.kernel_size[0] != conv.kernel_size[1] or conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1] \
                or conv.groups != 1 or conv.dilation != (1, 1) or isinstance(conv.padding, str):
            raise ValueError("exported convolutions must be square, ungrouped, undilated, with numeric padding")
        if conv.out_channels % num_group_elements:
            raise ValueError(f"{conv.out_channels} output channels are not a whole number of regular fields of size {num_group_elements}")
        self.conv = conv
        self.num_group_elements = num_group_elements
        self.lifting = lifting
        self.kernel_size, self.stride, self.padding = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        self.out_channels = conv.out_channels // num_group_elements          # fields, like the group convolutions

    @property
    def weights(self):
        return self.conv.weight

    @property
    def bias(self):
        return self.conv.bias                                                # per CHANNEL (fields x |G|)

    def expanded_weights(self) -> torch.Tensor:
        return self.conv.weight

    def mean_response_weights(self) -> torch.Tensor:
        w = self.conv.weight
        ver, hit = getattr(self, "_cached_weff", (-1, None))
        if ver != w._version or hit is None or hit.device != w.device:
            hit = w.detach().view(self.out_channels, self.num_group_elements, -1).double().sum(0)
            self._cached_weff = (w._version, hit)
        return hit

    def mean_bias_value(self):
        """Per-ELEMENT mean of the bias over the fields, (|G|,) fp64: an exported bias is constant within a regular field for an
        equivariant layer, but nothing here relies on that."""
        b = self.conv.bias
        if b is None:
            return 0.0
        ver, hit = getattr(self, "_cached_bias_mean", (-1, None))
        if ver != b._version or hit is None:
            m = b.detach().double().view(self.out_channels, self.num_group_elements).mean(0)
            hit = float(m[0].item()) if bool((m == m[0]).all().item()) else m
            self._cached_bias_mean = (b._version, hit)
        return hit

    def supports_linear_tail(self) -> bool:
        return self.stride == 1 and self.padding == 0 and self.kernel_size <= ops.MAX_WINDOW_K

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.conv(x)


class ESCNNEquivariantNetwork(nn.Module):
    def __init__(self, in_shape: tuple, out_channels: int, kernel_size: int, group_type: str = "rotation",
                 num_rotations: int = 4, num_layers: int = 1):
        super().__init__()
        self.in_channels = in_shape[0]
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.group_type = group_type
        self.num_rotations = num_rotations
        if group_type == "rotation":
            lift, conv = RotationEquivariantConvLift, RotationEquivariantConv
        elif group_type == "roto-reflection":
            lift, conv = RotoReflectionEquivariantConvLift, RotoReflectionEquivariantConv
        else:
            raise ValueError("group_type must be rotation or roto-reflection for now.")
        self.num_group_elements = num_rotations if group_type == "rotation" else 2 * num_rotations

        def block() -> List[nn.Module]:
            return [_InnerBatchNorm(out_channels, momentum=0.9), nn.ReLU(inplace=True), nn.Dropout(p=0.5)]

        # same module sequence as the reference ctor (:67-91): conv, [bn relu drop, conv] x (L-2), bn relu drop, conv
        mods: List[nn.Module] = [lift(self.in_channels, out_channels, kernel_size, num_rotations, device="cpu")]
        mods += block()
        for _ in range(num_layers - 2):
            mods.append(conv(out_channels, out_channels, kernel_size, num_rotations, device="cpu"))
            mods += block()
        mods.append(conv(out_channels, out_channels, kernel_size, num_rotations, device="cpu"))
        self.eqv_network = nn.Sequential(*mods)
        self._dense: Sequence = ()
        self._fold_cache: dict = {}

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """A checkpoint of the reference's e2cnn network (escnn_networks.py:48-91) cannot be loaded by key: e2cnn stores a flat
        coefficient vector per layer over its steerable basis (``...weights`` 1-D, ``...basisexpansion...`` / ``...filter``
        buffers), this network stores rotated filter banks.  Say so instead of failing on a shape mismatch."""
        own = prefix + "eqv_network."
        for k, v in state_dict.items():
            if k.startswith(own) and ("basisexpansion" in k or k.endswith(".filter") or k.endswith(".expanded_bias")
                                      or (k.endswith(".weights") and getattr(v, "dim", lambda: 2)() == 1)):
                raise RuntimeError(
                    f"state_dict key {k!r} comes from e2cnn's steerable-basis parameterisation (the reference's ESCNNEquivariantNetwork). "
                    "equiadapt_amd.ESCNNEquivariantNetwork parameterises its layers by rotated filter banks and cannot load it by key; "
                    "export the trained layers on the e2cnn side (R2Conv.export() -> nn.Conv2d, InnerBatchNorm.export() -> nn.BatchNorm2d) "
                    "and pass them to load_exported_dense(convs, norms).")
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def load_exported_dense(self, convs: Sequence[nn.Conv2d], norms: Sequence[nn.BatchNorm2d]) -> None:
        """Use exported dense layers (lists in network order) instead of the filter-bank parameterisation: the bridge for
        weights trained with e2cnn, whose steerable-basis parameters cannot be loaded by key (``R2Conv.export()`` ->
        ``nn.Conv2d`` over fields x |G| channels, ``InnerBatchNorm.export()`` -> ``nn.BatchNorm2d``; reference network:
        escnn_networks.py:48-91).  Inference runs through the same fast path as the filter-bank form (FFT / Winograd / MFMA
        lifting layer, folded batch-norms, linearised last layer); with autograd the plain modules are used."""
        n_conv = sum(1 for m in self.eqv_network if hasattr(m, "expanded_weights"))
        if len(convs) != n_conv or len(norms) != n_conv - 1:
            raise ValueError(f"expected {n_conv} convs and {n_conv - 1} norms")
        E = self.num_group_elements
        cin = self.in_channels
        for i, cv in enumerate(convs):
            if cv.in_channels != cin:
                raise ValueError(f"exported conv {i} takes {cv.in_channels} channels, the network feeds it {cin}")
            cin = cv.out_channels
        for i, bn in enumerate(norms):
            if bn.num_features != convs[i].out_channels:
                raise ValueError(f"exported norm {i} has {bn.num_features} features, conv {i} produces {convs[i].out_channels}")
        self._dense = (nn.ModuleList(_DenseConv(cv, E, i == 0) for i, cv in enumerate(convs)), nn.ModuleList(norms))
        self._fold_cache.clear()

    def export_dense(self):
        """(convs, norms): this network's layers in the exported dense form -- expanded filter banks as ``nn.Conv2d`` (bias per
        channel), InnerBatchNorm as ``nn.BatchNorm2d`` with every per-field quantity repeated |G| times -- i.e. what
        ``load_exported_dense`` takes.  Eval-mode equivalent of the filter-bank network."""
        E = self.num_group_elements
        convs, norms = [], []
        with torch.no_grad():
            for m in self.eqv_network:
                if hasattr(m, "expanded_weights"):
                    bank = m.expanded_weights().detach()
                    cv = nn.Conv2d(bank.shape[1], bank.shape[0], m.kernel_size, m.stride, m.padding, bias=m.bias is not None)
                    cv = cv.to(bank.device)
                    cv.weight.copy_(bank)
                    if m.bias is not None:
                        cv.bias.copy_(m.bias.detach().repeat_interleave(E))
                    convs.append(cv)
                elif isinstance(m, _InnerBatchNorm):
                    bn = nn.BatchNorm2d(m.num_features * E, eps=m.eps, momentum=m.momentum).to(m.weight.device)
                    for name in ("weight", "bias", "running_mean", "running_var"):
                        getattr(bn, name).copy_(getattr(m, name).repeat_interleave(E))
                    norms.append(bn.eval())
        return convs, norms

    def _layers(self):
        """(convs, norms) the forward paths iterate over: the exported dense layers if loaded, else the group convolutions."""
        if self._dense:
            return list(self._dense[0]), list(self._dense[1])
        mods = list(self.eqv_network)
        return [m for m in mods if hasattr(m, "expanded_weights")], [m for m in mods if isinstance(m, _InnerBatchNorm)]

    # -- inference fast path ------------------------------------------------------------------------------------
    def _folded(self, conv, bn):
        """Filter bank and bias of ``bn(conv(.))`` in eval mode: the per-field affine of the batch-norm is folded into
        the bank (one conv, no separate bias / normalisation passes over the feature map).  Cached per weight version."""
        key = (conv.weights._version, bn.weight._version, bn.bias._version, bn.running_mean._version,
               bn.running_var._version, str(conv.weights.device))
        hit = self._fold_cache.get(id(conv))
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        E = conv.num_group_elements
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)                    # per field (per channel: exported form)
        shift = bn.bias - bn.running_mean * scale
        b = conv.bias if conv.bias is not None else torch.zeros_like(scale)
        bias = b * scale + shift
        if not isinstance(conv, _DenseConv):
            scale, bias = scale.repeat_interleave(E), bias.repeat_interleave(E)
        bank = conv.expanded_weights() * scale[:, None, None, None]
        # channels-last: MIOpen's fp32 implicit-GEMM kernels are NHWC; keeping the bank (and the activations) in that
        # layout removes the NCHW<->NHWC transposes around every convolution
        bank, bias = bank.contiguous(memory_format=torch.channels_last), bias.contiguous()
        self._fold_cache[id(conv)] = (key, bank, bias)
        return bank, bias

    def _winograd_filters(self, conv, bn, bank, m):
        hit = self._fold_cache.get(("wino", m, id(conv)))
        key = self._fold_cache[id(conv)][0]
        if hit is None or hit[0] != key:
            hit = (key, winograd.transform_filters(bank, m))
            self._fold_cache[("wino", m, id(conv))] = hit
        return hit[1]

    def _fft_filters(self, conv, bank):
        hit = self._fold_cache.get(("fft", id(conv)))
        key = self._fold_cache[id(conv)][0]
        if hit is None or hit[0] != key:
            hit = (key, fftconv.spectra_for_k(bank))
            self._fold_cache[("fft", id(conv))] = hit
        return hit[1]

    def _lift_weights(self, conv, bank):
        from equiadapt_amd import ops

        hit = self._fold_cache.get(("lift", id(conv)))
        key = self._fold_cache[id(conv)][0]
        if hit is None or hit[0] != key:
            narrow = ops.lift_conv_supported(bank.shape[1], bank.shape[-2], bank.shape[-1], bank.shape[0])
            hit = (key, ops.pack_lift_weights(bank) if narrow else ops.pack_lift_weights_wide(bank))
            self._fold_cache[("lift", id(conv))] = hit
        return hit[1]

    def _forward_inference(self, x: torch.Tensor) -> torch.Tensor:
        """eval + no_grad: conv(+folded BN) -> ReLU ... -> [last conv + group mean as window sums]."""
        from equiadapt_amd import ops

        convs, norms = self._layers()
        nhwc = (self.out_channels * self.num_group_elements) % 4 == 0
        h = x.contiguous(memory_format=torch.channels_last) if nhwc else x
        pending = None  # bias of the previous layer whose (bias + ReLU) has not been applied to `h` yet
        for i, (conv, bn) in enumerate(zip(convs[:-1], norms)):
            bank, bias = self._folded(conv, bn)
            last_before_tail = i == len(convs) - 2
            is_5x5 = nhwc and not conv.lifting and conv.kernel_size == 5 and conv.stride == 1 and conv.padding == 0
            if is_5x5 and fftconv.applicable(h, bank.shape[1], bank.shape[0]):
                # 5x5 regular->regular layer whose output the 44-pixel FFT tiles fit: overlap-save FFT convolution, 2.5
                # multiplies per output (Winograd F(4x4,5x5): 4).  Same fusions as below: the previous layer's bias + ReLU
                # on the input loads, this layer's on the way out, window sums instead of the map in front of the tail.
                tail = convs[-1]
                Bf = self._fft_filters(conv, bank)
                if last_before_tail and tail.kernel_size in (3, 5) and tail.supports_linear_tail() and \
                        min(h.shape[-2:]) - 4 >= 2 * tail.kernel_size - 1:
                    S = fftconv.conv5x5(h, Bf, bias, True, pending, pending is not None, sums_k=tail.kernel_size)
                    return window_sums_to_activations(S, tail, h.shape[-2] - 4, h.shape[-1] - 4)
                h = fftconv.conv5x5(h, Bf, bias, True, pending, pending is not None)
                pending = None
                if last_before_tail:
                    return conv_then_group_pool(h, convs[-1])
                continue
            if isinstance(h, fftconv.GroupedMap):   # written for an FFT layer that did not take it after all
                h = h.to_channels_last()
            elif isinstance(h, fftconv.LiftedInput):
                h = h.materialize()
            if (nhwc and not conv.lifting and conv.kernel_size != 5 and conv.stride == 1 and conv.padding == 0
                    and h.is_contiguous(memory_format=torch.channels_last)
                    and fftconv.applicable_k(h.shape, bank.shape[1], bank.shape[0], conv.kernel_size, h.device)):
                # regular -> regular layer of another kernel size (the reference's tutorial network: k = 9): the same overlap-save
                # FFT convolution with 49 - k outputs per tile (eqa_fft48_*); previous layer's bias + ReLU on the input loads,
                # this layer's on the way out; the linearised tail reads the map through the window-sum kernel
                h = fftconv.conv_kxk(h, self._fft_filters(conv, bank), conv.kernel_size, bias, True, pending, pending is not None)
                pending = None
                if last_before_tail:
                    return conv_then_group_pool(h, convs[-1])
                continue
            use_wino = is_5x5 and winograd.applicable(h, bank.shape[1], bank.shape[0])
            if use_wino:
                # 5x5 regular->regular layer: Winograd F(m x m, 5x5), m = 4 where the size allows.  The previous layer's
                # bias + ReLU ride on its input loads, its own bias + ReLU on its output transform.
                tail = convs[-1]
                m = winograd.tile_for(h)
                if last_before_tail and not winograd.sums_applicable(h, tail.kernel_size, m) and \
                        winograd.sums_applicable(h, tail.kernel_size, 2):
                    m = 2
                if last_before_tail and winograd.sums_applicable(h, tail.kernel_size, m):
                    # the activation of this layer is consumed only through the next layer's window sums: emit those
                    # straight from the output transform, the feature map is never written
                    S = winograd.conv5x5(h, self._winograd_filters(conv, bn, bank, m), bias, relu=True, in_bias=pending,
                                         in_relu=pending is not None, sums_k=tail.kernel_size)
                    return window_sums_to_activations(S, tail, h.shape[-2] - 4, h.shape[-1] - 4)
                h = winograd.conv5x5(h, self._winograd_filters(conv, bn, bank, m), bias, relu=True,
                                     in_bias=pending, in_relu=pending is not None)
                pending = None
                if last_before_tail:
                    return conv_then_group_pool(h, convs[-1])
                continue
            if pending is not None:  # the next consumer cannot absorb it: apply in one fused pass
                ops.bias_relu_nhwc_(h, pending)
                pending = None
            k = conv.kernel_size
            if (nhwc and conv.lifting and conv.stride == 1 and conv.padding == 0 and os.environ.get("EQA_LIFT_MFMA", "1") != "0"
                    and ops.lift_conv_supported(bank.shape[1], k, k, bank.shape[0])):
                # lifting layer (RGB -> regular fields): hand-written fp32-MFMA implicit GEMM, bias + ReLU in its epilogue.
                # If the next layer is an FFT-convolved 5x5 layer, the map goes out channel-group-major, the layout that
                # layer's input transform reads in whole cache lines (it is consumed by nothing else).
                nxt = convs[i + 1] if i + 1 < len(convs) - 1 else None
                out_shape = (h.shape[0], bank.shape[0], h.shape[2] - k + 1, h.shape[3] - k + 1)
                if (nxt is not None and not nxt.lifting and nxt.kernel_size == 5 and nxt.stride == 1 and nxt.padding == 0
                        and fftconv.lift_fused_applicable(h.shape, bank.shape, nxt.out_channels * nxt.num_group_elements, h.device)):
                    # round 6: the layer does not run here at all -- the FFT layer behind it computes each tile of this map from
                    # its input patch inside its own forward transform (eqa_lift5_fft48k5_input)
                    pieces = None
                    if fftconv.LIFT_FFT_FORM == "bf16x3":     # the opt-in form's operand, cached per weight version like the folded bank
                        hit = self._fold_cache.get(("liftp", id(conv)))
                        key = self._fold_cache[id(conv)][0]
                        if hit is None or hit[0] != key:
                            hit = (key, fftconv.LiftedInput(h, bank, bias, True).pieces())
                            self._fold_cache[("liftp", id(conv))] = hit
                        pieces = hit[1]
                    hit = None
                    if fftconv.LIFT_FFT_FORM == "h2":         # the fp16 form's operand (and its scale: one host synchronisation per weight version)
                        hit = self._fold_cache.get(("lifth", id(conv)))
                        if hit is not None and hit[0] != self._fold_cache[id(conv)][0]:
                            hit = None
                    h = fftconv.LiftedInput(h, bank, bias, True, pieces, hit[1] if hit is not None else None)
                    if fftconv.LIFT_FFT_FORM == "h2" and hit is None:
                        self._fold_cache[("lifth", id(conv))] = (self._fold_cache[id(conv)][0], h.pieces_f16())
                    continue
                if (nxt is not None and not nxt.lifting and nxt.kernel_size == 5 and nxt.stride == 1 and nxt.padding == 0
                        and fftconv.grouped_applicable(out_shape, bank.shape[0], bank.shape[0], h.device)):
                    h = fftconv.GroupedMap(ops.lift_conv_grouped(h, self._lift_weights(conv, bank), bias, True, k, k))
                    continue
                h = ops.lift_conv_nhwc(h, self._lift_weights(conv, bank), bias, True, k, k)
                if last_before_tail:
                    return conv_then_group_pool(h, convs[-1])
                continue
            if (nhwc and conv.lifting and conv.stride == 1 and conv.padding == 0 and os.environ.get("EQA_LIFT_MFMA", "1") != "0"
                    and h.shape[-2] >= k and h.shape[-1] >= k and ops.lift_conv_wide_supported(bank.shape[1], k, k, bank.shape[0])):
                # the lifting filters the kernel above does not take (7 x 7 / 9 x 9 over RGB: the reference tutorial's k = 9; grayscale)
                h = ops.lift_conv_wide(h, self._lift_weights(conv, bank), bias, True, k, k)
                if last_before_tail:
                    return conv_then_group_pool(h, convs[-1])
                continue
            if last_before_tail:
                # bias + ReLU of this layer are applied inside the window-sum pass of the next (last) layer
                c = F.conv2d(h, bank)
                return conv_then_group_pool(c, convs[-1], shift=bias, relu=True)
            h = F.conv2d(h, bank)
            if nhwc and h.is_contiguous(memory_format=torch.channels_last):
                pending = bias                                    # deferred: fused into whatever reads h next
            else:
                h = torch.relu_(h + bias[None, :, None, None])
        raise AssertionError("unreachable: the network always has at least two convolutions")

    # -- training fast path ------------------------------------------------------------------------------------
    @staticmethod
    def _inner_bn(h: torch.Tensor, bn: nn.BatchNorm3d, E: int, conv_bias) -> torch.Tensor:
        """InnerBatchNorm on a channels-last (B, fields*E, H, W) map: statistics per field over (batch, group, space), fp64
        accumulation, running statistics updated like nn.BatchNorm3d (momentum, unbiased variance).  The convolution's own
        bias only shifts the mean (it cancels in the normalised output), so it enters the running mean and nowhere else."""
        Bn, C, H, W = h.shape
        Fd = C // E
        if bn.training or bn.running_mean is None:
            n = Bn * E * H * W
            s1 = h.sum(dim=(0, 2, 3), dtype=torch.float64).view(Fd, E).sum(1)
            s2 = (h * h).sum(dim=(0, 2, 3), dtype=torch.float64).view(Fd, E).sum(1)
            mean = s1 / n
            var = (s2 / n - mean * mean).clamp_min(0.0)
            with torch.no_grad():
                full_mean = mean if conv_bias is None else mean + conv_bias.double()
                update_running_stats(bn, full_mean, var * (n / max(n - 1, 1)))
            mean, var = mean.float(), var.float()
        else:
            mean = bn.running_mean if conv_bias is None else bn.running_mean - conv_bias
            var = bn.running_var
        scale = bn.weight * torch.rsqrt(var + bn.eps)
        shift = bn.bias - mean * scale
        return h * scale.repeat_interleave(E)[None, :, None, None] + shift.repeat_interleave(E)[None, :, None, None]

    def _forward_training(self, x: torch.Tensor) -> torch.Tensor:
        """Same layers as ``self.eqv_network`` + group pooling, with autograd, channels-last: 5x5 regular->regular
        convolutions through Winograd (``winograd.Conv5x5Function``: forward, d/dx and d/dfilters on the Winograd kernels),
        the last convolution + group mean as window sums + GEMV (``WindowSumsFunction``), batch-norm / ReLU / dropout as
        element-wise passes.  Measured on the headline net, B = 256: 353 ms per step through MIOpen -> see HISTORY.md."""
        mods = list(self.eqv_network)
        convs = [m for m in mods if hasattr(m, "expanded_weights")]
        norms = [m for m in mods if isinstance(m, _InnerBatchNorm)]
        drops = [m for m in mods if isinstance(m, nn.Dropout)]
        E = self.num_group_elements
        h = x.contiguous(memory_format=torch.channels_last)
        tail = convs[-1]
        S = None
        epilogue_stats = os.environ.get("EQA_TRAIN_FUSED_BN", "1") != "0" and os.environ.get("EQA_TRAIN_EPILOGUE_STATS", "1") != "0"
        for li, (conv, bn, drop) in enumerate(zip(convs[:-1], norms, drops)):
            bank = conv.expanded_weights()
            part = None          # fp64 partial sums of the block's batch statistics, when the convolution kernel took them
            if not conv.lifting and conv.kernel_size == 5 and winograd.applicable(h, bank.shape[1], bank.shape[0]):
                if (epilogue_stats and (bn.training or bn.running_mean is None) and winograd.Conv5x5Function.stats_supported(h, bank)):
                    h, part = winograd.Conv5x5Function.apply(h, bank, winograd.tile_for(h), True)
                else:
                    h = winograd.Conv5x5Function.apply(h, bank, winograd.tile_for(h))
            elif (not conv.lifting and conv.kernel_size != 5
                  and fftconv.applicable_k(h.shape, bank.shape[1], bank.shape[0], conv.kernel_size, h.device,
                                           h.is_contiguous(memory_format=torch.channels_last))):
                # kernel sizes Winograd F(m, 5) does not cover (the tutorial's k = 9): forward and both gradients as FFT convolutions
                h = fftconv.ConvKxKFunction.apply(h, bank)
            elif (conv.lifting and os.environ.get("EQA_LIFT_MFMA", "1") != "0"
                  and ops.lift_conv_supported(bank.shape[1], conv.kernel_size, conv.kernel_size, bank.shape[0])):
                # batch statistics of the norm behind the layer: taken in the convolution's epilogue where the kernel has that form
                if (epilogue_stats and (bn.training or bn.running_mean is None)
                        and ops.lift_conv_stats_supported(h.shape, conv.kernel_size, conv.kernel_size, bank.shape[0])):
                    h, part = LiftConvFunction.apply(h, bank, True)
                else:
                    h = LiftConvFunction.apply(h, bank)
            elif (conv.lifting and conv.stride == 1 and conv.padding == 0 and os.environ.get("EQA_LIFT_MFMA", "1") != "0"
                  and h.is_contiguous(memory_format=torch.channels_last) and h.shape[-2] >= conv.kernel_size and h.shape[-1] >= conv.kernel_size
                  and ops.lift_conv_wide_supported(bank.shape[1], conv.kernel_size, conv.kernel_size, bank.shape[0])):
                h = LiftConvFunction.apply(h, bank)          # forward: eqa_lift_conv_wide; filter gradient: the framework's
            else:
                h = F.conv2d(h, bank.contiguous(memory_format=torch.channels_last))
            fused = os.environ.get("EQA_TRAIN_FUSED_BN", "1") != "0" and h.is_contiguous(memory_format=torch.channels_last)
            kt = tail.kernel_size
            if (fused and li == len(convs) - 2 and os.environ.get("EQA_TRAIN_FUSED_TAIL", "1") != "0" and kt <= ops.MAX_WINDOW_K
                    and h.shape[-2] > 2 * (kt - 1) and h.shape[-1] > 2 * (kt - 1) and h.shape[0] <= 65535):
                # the last hidden block goes straight into the window sums of the linearised final layer: its output is never written
                S = InnerBnReluDropoutWindowSums.apply(h, bn.weight, bn.bias, bn, E, conv.bias, drop.p, drop.training, kt, part)
            elif fused:
                h = InnerBnReluDropout.apply(h, bn.weight, bn.bias, bn, E, conv.bias, drop.p, drop.training, part)
            else:  # op-by-op form of the same block (kept as the reference for the fused kernels' test)
                h = self._inner_bn(h, bn, E, conv.bias)
                h = F.dropout(torch.relu(h), drop.p, drop.training)
        k, O = tail.kernel_size, tail.out_channels
        H, W = h.shape[-2:]
        if S is None:
            S = WindowSumsFunction.apply(h, k)                                      # (B, C, k, k) fp64
        weff = tail.expanded_weights().view(O, E, -1).double().sum(0)               # (E, C*k*k), differentiable
        from equiadapt_amd.images.canonicalization_networks.pooling import WindowSumsLinearFn

        if S.is_cuda and S.dtype == torch.float64 and E <= 16 and os.environ.get("EQA_WS_TABLE_KERNEL", "1") != "0":
            act = WindowSumsLinearFn.apply(S.flatten(1), weff, 1.0 / float(O * (H - k + 1) * (W - k + 1)))     # (B, E) fp32
        else:
            act = S.flatten(1) @ weff.t() / float(O * (H - k + 1) * (W - k + 1))
        if tail.bias is not None:
            act = act + tail.bias.double().mean()
        # hidden-layer convolution biases cancel inside the batch-norms (zero gradient); keep them in the graph with that
        # exact zero so that DistributedDataParallel sees a gradient for every parameter
        hidden_bias = [c.bias.sum() for c in convs[:-1] if c.bias is not None]
        if hidden_bias:
            act = act + 0.0 * torch.stack(hidden_bias).sum().double()
        return act.float()

    def _training_fast_path_ok(self, x: torch.Tensor) -> bool:
        if self._dense or not x.is_cuda or x.dtype != torch.float32 or os.environ.get("EQA_TRAIN_FAST", "1") == "0":
            return False
        if (self.out_channels * self.num_group_elements) % 4 != 0:
            return False
        convs = [m for m in self.eqv_network if hasattr(m, "expanded_weights")]
        if len(convs) < 2 or any(c.stride != 1 or c.padding != 0 for c in convs) or not convs[-1].supports_linear_tail():
            return False
        hw = x.shape[-2] - (self.kernel_size - 1) * (len(convs) - 1)
        k = convs[-1].kernel_size
        return hw >= 2 * k - 1 and x.shape[-1] - (self.kernel_size - 1) * (len(convs) - 1) >= 2 * k - 1 and hw * hw <= 12288

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (self.training or torch.is_grad_enabled()) and self._training_fast_path_ok(x):
            return self._forward_training(x)
        if not self.training and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32:
            convs, norms = self._layers()
            shrink = (self.kernel_size - 1) * (len(convs) - 1)
            oh, ow = x.shape[-2] - shrink, x.shape[-1] - shrink
            # (the exported dense layers are not registered submodules, so .eval() does not reach them: a norm left in training
            # mode must take the module path with its batch statistics, not the folded running statistics)
            if (convs[-1].supports_linear_tail() and 0 < oh and 0 < ow and oh * ow <= 12288
                    and all(n.running_mean is not None and not n.training for n in norms)
                    and all(c.stride == 1 and c.padding == 0 and c.kernel_size == self.kernel_size for c in convs)):
                return self._forward_inference(x)
        if self._dense:
            convs, norms = self._dense
            for i, cv in enumerate(convs):
                x = cv(x)
                if i < len(norms):
                    x = F.relu(norms[i](x))
            fm = x.reshape(x.shape[0], self.out_channels, self.num_group_elements, x.shape[-2], x.shape[-1])
        else:
            fm = self.eqv_network(x)
        return group_pool(fm)
