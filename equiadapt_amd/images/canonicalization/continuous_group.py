"""Continuous-group (SO(2)) image canonicalizers on MI355X.

API mirror of ``equiadapt/images/canonicalization/continuous_group.py``: same class names,
constructor arguments, attributes and ``canonicalization_info_dict`` keys.  The group element is a 2x2 matrix per image;
``canonicalize`` is the same fused pad / resample / crop kernel as the discrete path (``eqa_group_action_fwd``), fed with one
affine row PER SAMPLE instead of an index into a group table, and its backward (``eqa_group_action_bwd_theta``) returns
d/d(matrix) so the task loss trains the canonicalization network through the warp like autograd through
``K.geometry.warp_affine`` does in the reference.

Reference behaviour that is kept, not repaired (so that a user switching over sees the same thing):
* ``canonicalize`` negates the off-diagonal entries of the predicted matrices (:178, in place there): afterwards
  ``canonicalization_info_dict["group_element"]["rotation"]`` and ``["group_element_matrix_representation"]`` hold the INVERSE
  rotations;
* the warp is centred on ``(Hp // 2, Wp // 2)`` with the height in the x role (:192);
* ``group_type == "roto-reflection"`` cannot run in the reference: ``gram_schmidt`` (common/utils.py:44) indexes a third
  vector of the (B, 2, 2) output and raises IndexError.  Here it raises NotImplementedError naming that line;
* ``invert_canonicalization`` goes through ``get_action_on_image_features`` with the class's EMPTY ``group_info_dict`` (:86), so
  it raises ``KeyError('num_rotations')`` there and here;
* ``OptimizedSteerableImageCanonicalization.group_augment`` builds its matrices with
  ``torch.stack((cos, -sin, sin, cos)).reshape(-1, 2, 2)`` (:365-367), which for a batch of more than one image mixes the
  samples' sines and cosines; the augmented images and the "ground-truth" matrices of the loss follow that.
"""
import math
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn.functional as F

from equiadapt_amd import ops
from equiadapt_amd.common.basecanonicalization import ContinuousGroupCanonicalization
from equiadapt_amd.images import geometry
from equiadapt_amd.images.canonicalization.discrete_group import DiscreteGroupImageCanonicalization, _as_size
from equiadapt_amd.images.transforms import CenterCrop, EdgePad, Resize
from equiadapt_amd.images.utils import get_action_on_image_features


class _WarpAffineFn(torch.autograd.Function):
    """y = crop(warp(pad(x))) with one affine row per sample; backward: d/dx (scatter) and d/dtheta (6 sums per image)."""

    @staticmethod
    def forward(ctx, x, theta, pad, out_hw, top_left):
        idx = torch.arange(x.shape[0], dtype=torch.int32, device=x.device)
        theta = theta.contiguous()
        ctx.save_for_backward(x, theta, idx)
        ctx.pad, ctx.top_left = pad, top_left
        return ops.group_action(x, idx, theta, None, None, pad, out_hw, top_left)

    @staticmethod
    def backward(ctx, grad_y):
        x, theta, idx = ctx.saved_tensors
        need_x, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gx = gt = None
        if need_x or need_t:
            gx, gt = ops.group_action_bwd(x, grad_y.contiguous(), idx, theta, None, None, ctx.pad, ctx.top_left, need_x,
                                          False, want_theta=need_t)
            if not need_t:
                gt = None
        return gx, gt, None, None, None


def _get(hyperparams: Any, name: str):
    return hyperparams[name] if isinstance(hyperparams, dict) else getattr(hyperparams, name)


class ContinuousGroupImageCanonicalization(ContinuousGroupCanonicalization):
    """Base of the continuous-group image canonicalizers (reference: continuous_group.py:15-230)."""

    def __init__(self, canonicalization_network: torch.nn.Module, canonicalization_hyperparams: Any, in_shape: tuple):
        super().__init__(canonicalization_network)
        assert len(in_shape) == 3, "Input shape should be in the format (channels, height, width)"
        self.in_shape = tuple(int(s) for s in in_shape)
        is_grayscale = self.in_shape[0] == 1
        # kept as attributes for API parity; canonicalize() fuses pad + warp + crop instead of calling them
        self.pad_size = 0 if is_grayscale else math.ceil(self.in_shape[-1] * 0.5)
        self.pad = torch.nn.Identity() if is_grayscale else EdgePad(self.pad_size)
        self.crop = torch.nn.Identity() if is_grayscale else CenterCrop((self.in_shape[-2], self.in_shape[-1]))
        self.crop_canonization = (
            torch.nn.Identity() if is_grayscale else CenterCrop((
                math.ceil(self.in_shape[-2] * _get(canonicalization_hyperparams, "input_crop_ratio")),
                math.ceil(self.in_shape[-1] * _get(canonicalization_hyperparams, "input_crop_ratio"))))
        )
        self.resize_canonization = (
            torch.nn.Identity() if is_grayscale
            else Resize(size=_as_size(_get(canonicalization_hyperparams, "resize_shape")))
        )
        self.group_info_dict: Dict[str, Any] = {}
        self._consts: Dict[Any, Any] = {}

    def get_groupelement(self, x: torch.Tensor) -> dict:
        raise NotImplementedError("get_groupelement method is not implemented")

    # crop + antialiased resize, fused into one kernel without autograd -- shared with the discrete classes
    transformations_before_canonicalization_network_forward = (
        DiscreteGroupImageCanonicalization.transformations_before_canonicalization_network_forward)

    def get_group_from_out_vectors(self, out_vectors: torch.Tensor) -> Tuple[dict, torch.Tensor]:
        """Network output (B, n_vectors, 2) -> ({"rotation": (B, 2, 2)}, matrix representation) (reference :115-160)."""
        if self.group_type == "roto-reflection":
            raise NotImplementedError(
                "roto-reflection is not runnable in the reference either: gram_schmidt (equiadapt/common/utils.py:44) reads "
                "a third vector of the (B, 2, 2) network output and raises IndexError")
        rotation_matrices = self.get_rotation_matrix_from_vector(out_vectors[:, 0])
        return {"rotation": rotation_matrices}, rotation_matrices

    def canonicalize(self, x: torch.Tensor, targets: Optional[List] = None, **kwargs: Any
                     ) -> Union[torch.Tensor, Tuple[torch.Tensor, List]]:
        """Rotate every image by the inverse of its predicted rotation (reference :162-210)."""
        self.device = x.device
        group_element_dict = self.get_groupelement(x)
        rotation_matrices = group_element_dict["rotation"]
        # the reference negates the off-diagonal IN PLACE (inverse rotation); same observable state, out of place
        sign = torch.tensor([[1.0, -1.0], [-1.0, 1.0]], dtype=rotation_matrices.dtype, device=rotation_matrices.device)
        inverse = rotation_matrices * sign
        info = self.canonicalization_info_dict
        if info.get("group_element_matrix_representation") is rotation_matrices:
            info["group_element_matrix_representation"] = inverse
        group_element_dict["rotation"] = inverse
        if "reflection" in group_element_dict:
            r = group_element_dict["reflection"]
            x = (1 - r) * x + r * x.flip(-1)
        H, W = x.shape[-2:]
        pad = self.pad_size
        Hp, Wp = H + 2 * pad, W + 2 * pad
        alpha, beta = inverse[:, 0, 0], inverse[:, 0, 1]
        cx, cy = Hp // 2, Wp // 2
        affine_part = torch.stack([(1 - alpha) * cx - beta * cy, beta * cx + (1 - alpha) * cy], dim=1)
        M = torch.cat([inverse, affine_part.unsqueeze(-1)], dim=-1)
        theta = geometry.warp_affine_theta(M, (Hp, Wp)).to(torch.float32)
        return _WarpAffineFn.apply(x, theta, pad, (H, W), (pad, pad))

    def invert_canonicalization(self, x_canonicalized_out: torch.Tensor, **kwargs: Any) -> torch.Tensor:
        """Reference :212-230: delegates to get_action_on_image_features with this class's (empty) ``group_info_dict``."""
        induced_rep_type = kwargs.get("induced_rep_type", "vector")
        return get_action_on_image_features(
            feature_map=x_canonicalized_out,
            group_info_dict=self.group_info_dict,
            group_element_dict=self.canonicalization_info_dict["group_element"],
            induced_rep_type=induced_rep_type,
        )


def _rotation_matrix_from_vector(vectors: torch.Tensor) -> torch.Tensor:
    """(B, 2) -> (B, 2, 2): rows v1 = v / |v|, v2 = (-v1_y, v1_x) (reference :246-261, :331-346)."""
    v1 = vectors / torch.norm(vectors, dim=1, keepdim=True)
    v2 = torch.stack([-v1[:, 1], v1[:, 0]], dim=1)
    return torch.stack([v1, v2], dim=1)


class SteerableImageCanonicalization(ContinuousGroupImageCanonicalization):
    """Canonicalizer driven by a rotation-EQUIVARIANT (steerable) network that outputs 2-vectors (reference :233-311)."""

    def __init__(self, canonicalization_network: torch.nn.Module, canonicalization_hyperparams: Any, in_shape: tuple):
        super().__init__(canonicalization_network, canonicalization_hyperparams, in_shape)
        self.group_type = canonicalization_network.group_type

    def get_rotation_matrix_from_vector(self, vectors: torch.Tensor) -> torch.Tensor:
        return _rotation_matrix_from_vector(vectors)

    def get_groupelement(self, x: torch.Tensor) -> dict:
        x = self.transformations_before_canonicalization_network_forward(x)
        out_vectors = self.canonicalization_network(x)
        if not hasattr(self, "canonicalization_info_dict"):
            self.canonicalization_info_dict = {}
        group_element_dict, representation = self.get_group_from_out_vectors(out_vectors)
        self.canonicalization_info_dict["group_element_matrix_representation"] = representation
        self.canonicalization_info_dict["group_element"] = group_element_dict
        return group_element_dict


class OptimizedSteerableImageCanonicalization(ContinuousGroupImageCanonicalization):
    """Canonicalizer driven by ANY network: equivariance is learnt by regressing the matrices of randomly transformed
    copies (reference :314-497)."""

    def __init__(self, canonicalization_network: torch.nn.Module, canonicalization_hyperparams: Any, in_shape: tuple):
        super().__init__(canonicalization_network, canonicalization_hyperparams, in_shape)
        self.group_type = _get(canonicalization_hyperparams, "group_type")

    def get_rotation_matrix_from_vector(self, vectors: torch.Tensor) -> torch.Tensor:
        return _rotation_matrix_from_vector(vectors)

    def group_augment(self, x: torch.Tensor, angles: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Random rotations of the batch and their matrices (reference :348-398).  ``angles`` (radians, (B,)) replaces the
        ``torch.rand`` draw, for tests."""
        if self.group_type == "roto-reflection":
            raise NotImplementedError("roto-reflection: see get_group_from_out_vectors")
        B = x.shape[0]
        if angles is None:
            angles = torch.rand(B, device=x.device) * 2 * torch.pi
        cos_a, sin_a = torch.cos(angles), torch.sin(angles)
        rotation_matrices = torch.zeros(B, 2, 3, device=x.device)
        # (4, B) reshaped to (B, 2, 2): for B > 1 this mixes the samples -- reference behaviour, kept (:365-367)
        rotation_matrices[:, :2, :2] = torch.stack((cos_a, -sin_a, sin_a, cos_a)).reshape(-1, 2, 2)
        H, W = x.shape[-2:]
        pad = self.pad_size
        theta = geometry.affine_grid_theta_half_pixel(rotation_matrices, (H + 2 * pad, W + 2 * pad))
        idx = torch.arange(B, dtype=torch.int32, device=x.device)
        with torch.no_grad():
            augmented = ops.group_action(x, idx, theta.contiguous(), None, None, pad, (H, W), (pad, pad))
        sign = torch.tensor([[1.0, -1.0], [-1.0, 1.0]], device=x.device)
        return augmented, rotation_matrices[:, :, :2] * sign

    def get_groupelement(self, x: torch.Tensor) -> dict:
        self.device = x.device
        batch_size = x.shape[0]
        x_augmented, gt = self.group_augment(x)
        x_all = torch.cat([x, x_augmented], dim=0)
        x_all = self.transformations_before_canonicalization_network_forward(x_all)
        out_vectors_all = self.canonicalization_network(x_all).reshape(2 * batch_size, -1, 2)
        out_vectors, out_vectors_augmented = out_vectors_all.chunk(2, dim=0)
        if not hasattr(self, "canonicalization_info_dict"):
            self.canonicalization_info_dict = {}
        group_element_dict, representation = self.get_group_from_out_vectors(out_vectors)
        self.canonicalization_info_dict["group_element_matrix_representation"] = representation
        self.canonicalization_info_dict["group_element"] = group_element_dict
        _, representation_augmented = self.get_group_from_out_vectors(out_vectors_augmented)
        self.canonicalization_info_dict["group_element_matrix_representation_augmented"] = representation_augmented
        self.canonicalization_info_dict["group_element_matrix_representation_augmented_gt"] = gt
        return group_element_dict

    def get_optimization_specific_loss(self) -> torch.Tensor:
        """MSE between the matrices predicted for the augmented copies and the augmentation's own (reference :472-497)."""
        return F.mse_loss(
            self.canonicalization_info_dict["group_element_matrix_representation_augmented"],
            self.canonicalization_info_dict["group_element_matrix_representation_augmented_gt"],
        )
