"""Group pooling: (B, C, G, H', W') -> (B, G) activations on the HIP streaming-reduction kernel."""
import torch

from equiadapt_amd import ops


class _GroupPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feature_map: torch.Tensor) -> torch.Tensor:
        ctx.shape = tuple(feature_map.shape)
        act, _ = ops.group_pool_argmax(feature_map, want_index=False)
        return act

    @staticmethod
    def backward(ctx, grad_act: torch.Tensor):
        B, C, G, H, W = ctx.shape
        # d mean / d feature = 1 / (C*H*W), broadcast: left to autograd consumers as an expanded view
        return (grad_act / float(C * H * W))[:, None, :, None, None].expand(B, C, G, H, W)


def group_pool(feature_map: torch.Tensor) -> torch.Tensor:
    """``torch.mean(feature_map, dim=(1, 3, 4))`` (reference: escnn_networks.py:115,
    custom_equivariant_networks.py:91) as one pass over the map (eqa_group_pool_argmax; fp32 -- a module converted to
    another dtype, e.g. ``.double()`` for a reference computation, takes torch's reduction)."""
    if feature_map.dtype != torch.float32:
        return torch.mean(feature_map, dim=(1, 3, 4))
    return _GroupPoolFn.apply(feature_map)


def conv_then_group_pool(h: torch.Tensor, layer, scale=None, shift=None, relu: bool = False) -> torch.Tensor:
    """``group_pool(layer(act(h)))`` WITHOUT running the convolution (inference fast path; exact up to fp rounding).

    The mean over (fields, space) of a convolution is linear in its input, so
        act[b, g] = sum_{c,u,v} Weff[g,c,u,v] * S[b,c,u,v] / count + mean(bias),
    with Weff = the filter bank summed over output fields and S the k*k shifted-window sums of each input plane
    (``eqa_window_sums``, one pass over ``h``; ``act(t) = [relu](scale*t + shift)`` per channel folds the previous
    layer's bias / eval-mode batch-norm / ReLU into that pass).  ``layer`` is a group conv with stride 1, padding 0.
    """
    k, E, O = layer.kernel_size, layer.num_group_elements, layer.out_channels
    B, _, H, W = h.shape
    S = ops.window_sums(h, k, scale, shift, relu)                       # (B, Cin, k, k) fp64
    return window_sums_to_activations(S, layer, H, W)


def window_sums_to_activations(S: torch.Tensor, layer, H: int, W: int) -> torch.Tensor:
    """The GEMV half of ``conv_then_group_pool``: (B, Cin, k, k) fp64 window sums of the (H, W) input -> (B, G)."""
    k, O = layer.kernel_size, layer.out_channels
    weff = layer.mean_response_weights()                                # (E, Cin*k*k) fp64
    scale = 1.0 / float(O * (H - k + 1) * (W - k + 1))
    if S.is_cuda and weff.shape[0] <= 16 and not torch.is_grad_enabled():
        from equiadapt_amd import ops

        return ops.window_sums_gemv(S.flatten(1), weff, scale, layer.mean_bias_value())
    act = S.flatten(1) @ weff.t() * scale
    if layer.bias is not None:
        mb = layer.mean_bias_value()                                    # a float, or one value per element (exported dense form)
        act = act + (mb if isinstance(mb, float) else mb.to(act.device))
    return act.float()


class WindowSumsLinearFn(torch.autograd.Function):
    """act = scale * S . weff^T, the GEMV of the linearised last layer, with its backward (dS = scale * dact . weff,
    dweff = scale * dact^T . S) on the library's kernels: eqa_window_sums_gemv / eqa_window_sums_gemv_bwd.  S:(B,K) fp64,
    weff:(E <= 16, K) fp64 -> (B,E) fp32.  (Through torch these are three fp64 library GEMMs with an inner or outer dimension of 4-8.)"""

    @staticmethod
    def forward(ctx, S, weff, scale):
        from equiadapt_amd import ops

        S, weff = S.contiguous(), weff.contiguous()
        ctx.save_for_backward(S, weff)
        ctx.scale = scale
        return ops.window_sums_gemv(S, weff, scale, 0.0)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dact):
        from equiadapt_amd import ops

        S, weff = ctx.saved_tensors
        dS, dW = ops.window_sums_gemv_bwd(dact.float().contiguous(), weff, S, ctx.scale, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dS, dW, None


class WindowSumsFunction(torch.autograd.Function):
    """S[b,c,u,v] = sum over the (H-k+1, W-k+1) window at (u, v) of x[b,c] (fp64), with its backward: a pixel (y, x) receives
    the sum of dS over the windows that contain it.  Rows fall into 2(k-1)+1 classes (the k-1 top rows, the interior, the k-1
    bottom rows), columns likewise, so the gradient is a (2k-1) x (2k-1) table per (b, c), expanded to the map by `eqa_window_sums_bwd_expand_nhwc`.
    Training counterpart of `conv_then_group_pool` (the last convolution + group mean are linear in x)."""

    @staticmethod
    def forward(ctx, x, k):
        from equiadapt_amd import ops

        ctx.k, ctx.shape = k, x.shape
        ctx.channels_last = x.is_contiguous(memory_format=torch.channels_last)
        return ops.window_sums(x, k)

    @staticmethod
    def backward(ctx, dS):
        k = ctx.k
        B, C, H, W = ctx.shape
        nb = k - 1
        dev = dS.device

        def classes(n):
            rep = torch.cat([torch.arange(nb), torch.tensor([nb]), torch.arange(n - nb, n)]).to(dev)       # one row per class
            u = torch.arange(k, device=dev)
            mask = ((u[None, :] <= rep[:, None]) & (rep[:, None] <= u[None, :] + (n - k))).to(dS.dtype)     # (2nb+1, k)
            i = torch.arange(n, device=dev)
            idx = torch.where(i < nb, i, torch.where(i >= n - nb, i - (n - nb) + nb + 1, torch.full_like(i, nb)))
            return mask, idx

        rm, ty = classes(H)
        cm, tx = classes(W)
        if dS.is_cuda and dS.dtype == torch.float64 and B <= 65535 and H >= 2 * nb + 1 and W >= 2 * nb + 1:
            from equiadapt_amd import ops

            table = ops.window_grad_table(dS.contiguous(), H, W)                     # (B, 2nb+1, 2nb+1, C), channels last
        else:
            table = torch.einsum("tu,bcuv,sv->btsc", rm, dS, cm).float().contiguous()
        if table.is_cuda and C % 4 == 0:
            from equiadapt_amd import _lib, ops

            g = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                st = _lib.load().eqa_window_sums_bwd_expand_nhwc(table.data_ptr(), g.data_ptr(), B, H, W, C, k, ops._stream())
            _lib.check(st, "eqa_window_sums_bwd_expand_nhwc")
        else:
            g = table[:, ty][:, :, tx]                                           # (B, H, W, C)
        g = g.permute(0, 3, 1, 2)                                                # NCHW view of channels-last memory
        if not ctx.channels_last:
            g = g.contiguous()
        return g, None
