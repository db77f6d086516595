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
from equiadapt_amd.images.canonicalization_networks.pooling import group_pool


class _InnerBatchNorm(nn.BatchNorm3d):
    """Per-field batch norm over (batch, group, space) of a (B, fields, G, H, W) map."""


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

    def load_exported_dense(self, convs: Sequence[nn.Conv2d], norms: Sequence[nn.BatchNorm2d]) -> None:
        """Use e2cnn-exported dense layers (Conv2d / BatchNorm2d lists in network order) instead of the
        filter-bank parameterisation.  Inference only."""
        n_conv = sum(1 for m in self.eqv_network if hasattr(m, "expanded_weights"))
        if len(convs) != n_conv or len(norms) != n_conv - 1:
            raise ValueError(f"expected {n_conv} convs and {n_conv - 1} norms")
        self._dense = (nn.ModuleList(convs), nn.ModuleList(norms))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
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
