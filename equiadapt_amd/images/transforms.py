"""Small tensor transforms with the semantics of the torchvision 0.17 classes the reference builds
(``transforms.Pad(p, "edge")``, ``CenterCrop``, ``Resize``): kept as attributes (`pad`, `crop`,
`crop_canonization`, `resize_canonization`) because callers and the reference's tests read them
(tests/images/canonicalization/test_continuous_group.py:63-64).  The canonicalize hot path does NOT
run pad/crop as separate passes -- they are fused into the resampling kernel.
"""
from typing import Sequence, Tuple, Union

import torch
import torch.nn.functional as F

from equiadapt_amd.images.geometry import center_crop_offset


class EdgePad(torch.nn.Module):
    """Replicate-pad all four sides by ``padding`` pixels."""

    def __init__(self, padding: int):
        super().__init__()
        self.padding = int(padding)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        p = self.padding
        return F.pad(x, (p, p, p, p), mode="replicate")

    def extra_repr(self) -> str:
        return f"padding={self.padding}, padding_mode=edge"


class CenterCrop(torch.nn.Module):
    """Crop the centre ``size`` window; offset = int(round((full - crop) / 2))."""

    def __init__(self, size: Union[int, Sequence[int]]):
        super().__init__()
        self.size: Tuple[int, int] = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        H, W = x.shape[-2:]
        h, w = self.size
        if h > H or w > W:
            raise ValueError(f"CenterCrop{self.size} larger than input {(H, W)} is not supported")
        top, left = center_crop_offset(H, h), center_crop_offset(W, w)
        return x[..., top : top + h, left : left + w]

    def extra_repr(self) -> str:
        return f"size={self.size}"


def resized_output_size(hw: Tuple[int, int], size: Union[int, Sequence[int]]) -> Tuple[int, int]:
    """int -> shorter edge becomes ``size`` (long edge int(size*long/short)); pair -> exact (h, w)."""
    h, w = hw
    if isinstance(size, int) or len(size) == 1:
        s = int(size if isinstance(size, int) else size[0])
        short, long = (w, h) if w <= h else (h, w)
        new_long = int(s * long / short)
        return (new_long, s) if w <= h else (s, new_long)
    return int(size[0]), int(size[1])


class Resize(torch.nn.Module):
    """Bilinear, align_corners=False, antialiased (torchvision 0.17 default for tensors)."""

    def __init__(self, size: Union[int, Sequence[int]], antialias: bool = True):
        super().__init__()
        self.size = size if isinstance(size, int) else tuple(int(s) for s in size)
        self.antialias = antialias

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        oh, ow = resized_output_size(tuple(x.shape[-2:]), self.size)
        return F.interpolate(x, size=[oh, ow], mode="bilinear", align_corners=False, antialias=self.antialias)

    def extra_repr(self) -> str:
        return f"size={self.size}, antialias={self.antialias}"
