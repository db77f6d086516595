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
    custom_equivariant_networks.py:91) as one pass over the map (eqa_group_pool_argmax)."""
    return _GroupPoolFn.apply(feature_map)
