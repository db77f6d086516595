"""Discrete-group (C_n / D_n) image canonicalizers on MI355X.

API mirror of ``equiadapt/images/canonicalization/discrete_group.py``: same class
names, constructor arguments (network, hyperparams read BY ATTRIBUTE, in_shape), attributes and state.
What differs is underneath:

* ``canonicalize``: the reference pads to (2H x 2W), blends with an h-flipped copy, resamples the whole
  padded frame and crops (about 8 MB of traffic per 224x224x3 image).  Here ONE kernel
  (``eqa_canon_transform_fwd``) computes only the H x W pixels that are kept, reading the source through
  an edge-clamp (== the replicate pad) with the flip folded into the addressing: 1.2 MB per image.
* the orientation index never leaves the device (no ``.item()``), so a whole step is graph-capturable.
* ``invert_canonicalization`` is one kernel as well (rotate + flip + regular-representation roll).
"""
import math
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn.functional as F

from equiadapt_amd import ops
from equiadapt_amd.common.basecanonicalization import DiscreteGroupCanonicalization
from equiadapt_amd.images import geometry
from equiadapt_amd.images.transforms import CenterCrop, EdgePad, Resize
from equiadapt_amd.images.utils import (
    canonicalize_masks,
    device_tables,
    canonicalize_boxes,
    flip_boxes,
    flip_masks,
    get_action_on_image_features,
    rotate_boxes,
    rotate_masks,
)



def _get(hyperparams: Any, name: str):
    """Hyper-parameters by attribute (SimpleNamespace / dataclass / DictConfig, as the reference reads them) or by key (a plain
    mapping: the reference's own tests build a DictConfig from one)."""
    if isinstance(hyperparams, dict):
        return hyperparams[name]
    return getattr(hyperparams, name)

class _CanonTransformFn(torch.autograd.Function):
    """y = crop(rotate(flip?(pad(x)), -rotation)).

    Forward: the fused HIP kernel, driven by the int32 element index (``rotation`` / ``reflection`` are the
    reference's straight-through tensors and only matter for autograd).  Backward (eqa_group_action_bwd):
    d/d rotation -- the path through which the task loss trains the canonicalizer in the reference (autograd
    through kornia's grid, discrete_group.py:213) --, d/d x, and d/d reflection = <g, T(hflip x) - T(x)>.
    """

    @staticmethod
    def forward(ctx, x, rotation, reflection, gidx, theta, flags, pad, num_rotations):
        ctx.save_for_backward(x, gidx, theta, flags if flags is not None else torch.empty(0))
        ctx.pad, ctx.num_rotations, ctx.has_flags = pad, num_rotations, flags is not None
        return ops.canon_transform(x, gidx, theta, flags, pad)

    @staticmethod
    def backward(ctx, grad_y):
        x, gidx, theta, flags = ctx.saved_tensors
        flags = flags if ctx.has_flags else None
        pad, N = ctx.pad, ctx.num_rotations
        need_x, need_rot, need_ref = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        grad_y = grad_y.contiguous()
        gx = g_rot = g_ref = None
        if need_x or need_rot:
            gx, g_ang = ops.group_action_bwd(x, grad_y, gidx, theta, flags, None, pad, (pad, pad), need_x, need_rot)
            if need_rot:
                g_rot = -g_ang  # the kernel differentiates w.r.t. the rotate() angle, which is -rotation here
        if need_ref:
            ridx = gidx % N
            y0 = ops.canon_transform(x, ridx, theta, flags, pad)
            y1 = ops.canon_transform(x, ridx + N, theta, flags, pad)
            g_ref = (grad_y * (y1 - y0)).sum(dim=(1, 2, 3))
        return gx, g_rot, g_ref, None, None, None, None, None


class DiscreteGroupImageCanonicalization(DiscreteGroupCanonicalization):
    """Base of the discrete-group image canonicalizers (reference: discrete_group.py:20-259)."""

    def __init__(self, canonicalization_network: torch.nn.Module, canonicalization_hyperparams: Any, in_shape: tuple):
        super().__init__(canonicalization_network)
        self.beta = _get(canonicalization_hyperparams, "beta")
        assert len(in_shape) == 3, "Input shape should be in the format (channels, height, width)"
        self.in_shape = tuple(int(s) for s in in_shape)
        is_grayscale = self.in_shape[0] == 1
        # kept as attributes for API parity; canonicalize() fuses pad+rotate+crop instead of calling them
        self.pad_size = 0 if is_grayscale else math.ceil(self.in_shape[-1] * 0.5)
        self.pad = torch.nn.Identity() if is_grayscale else EdgePad(self.pad_size)
        self.crop = torch.nn.Identity() if is_grayscale else CenterCrop((self.in_shape[-2], self.in_shape[-1]))
        self.crop_canonization = (
            torch.nn.Identity() if is_grayscale else CenterCrop((
                math.ceil(self.in_shape[-2] * _get(canonicalization_hyperparams, "input_crop_ratio")),
                math.ceil(self.in_shape[-1] * _get(canonicalization_hyperparams, "input_crop_ratio"))))
        )
        self.resize_canonization = (
            torch.nn.Identity() if is_grayscale else Resize(size=_as_size(_get(canonicalization_hyperparams, "resize_shape")))
        )
        self._consts: Dict[str, torch.Tensor] = {}

    # -- group bookkeeping -------------------------------------------------------------------------

    def _group_constants(self, device: torch.device) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """Per-element rotation angles (deg) and reflection indicators, cached on the device."""
        key = f"{device}:{self.num_rotations}:{self.group_type}"
        hit = self._consts.get(key)
        if hit is None:
            angles = geometry.group_angles(self.num_rotations)
            if self.group_type == "roto-reflection":
                rot = torch.cat([angles, angles], dim=0)
                ref = torch.cat([torch.zeros(self.num_rotations), torch.ones(self.num_rotations)], dim=0)
                hit = (rot.to(device), ref.to(device))
            else:
                hit = (angles.to(device), None)
            self._consts[key] = hit
        return hit

    def groupactivations_to_groupelement(self, group_activations: torch.Tensor,
                                         group_index: Optional[torch.Tensor] = None) -> dict:
        """(B, G) activations -> {"rotation" (deg)[, "reflection"], "group_index"} (reference :94-135).

        ``rotation`` / ``reflection`` carry the straight-through gradient in training mode exactly like the
        reference; ``group_index`` (int32, device) is what the kernels consume.
        """
        if group_index is None:
            group_index = self.group_index(group_activations)
        rot_comp, ref_comp = self._group_constants(group_activations.device)
        if not self.training and self.gradient_trick == "straight_through":
            # eval mode: the one-hot is the hard one, so sum(onehot * table) IS table[index] (one term, the rest exact zeros):
            # two lookups instead of one_hot / multiply / reduce per component (the step is launch-bound at small batches)
            element = {"rotation": torch.index_select(rot_comp, 0, group_index)}
            if ref_comp is not None:
                element["reflection"] = torch.index_select(ref_comp, 0, group_index)
            element["group_index"] = group_index
            return element
        onehot = self.groupactivations_to_groupelementonehot(group_activations, group_index)
        if self.gradient_trick != "straight_through":
            # gumbel_softmax: the one-hot is a random SAMPLE, not the argmax.  The reference applies the sampled element to
            # everything (rotation / reflection are sums over this one-hot, discrete_group.py:121-133), so the fused
            # kernels must consume the sampled index as well -- images, masks, boxes and the invert all see one element.
            group_index = onehot.detach().argmax(dim=-1).to(torch.int32)
        element = {"rotation": torch.sum(onehot * rot_comp, dim=-1)}
        if ref_comp is not None:
            element["reflection"] = torch.sum(onehot * ref_comp, dim=-1)
        element["group_index"] = group_index
        return element

    def get_group_activations(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError(
            "get_group_activations is not implemented for the DiscreteGroupImageCanonicalization class")

    def get_groupelement(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Image batch -> group element per image; records activations/element in the info dict (:152-172)."""
        group_activations = self.get_group_activations(x)
        group_index = self.group_index(group_activations)
        element = self.groupactivations_to_groupelement(group_activations, group_index)
        if not hasattr(self, "canonicalization_info_dict"):
            self.canonicalization_info_dict = {}
        # "group_index" stays out of the element dict callers iterate over (they test `"reflection" in ...`)
        self.canonicalization_info_dict["group_index"] = element.pop("group_index")   # the element applied (sampled under gumbel)
        self.canonicalization_info_dict["argmax_index"] = group_index                 # argmax of the activations (identity metric)
        self.canonicalization_info_dict["group_element"] = element
        self.canonicalization_info_dict["group_activations"] = group_activations
        return element

    def transformations_before_canonicalization_network_forward(self, x: torch.Tensor) -> torch.Tensor:
        """Centre crop by ``input_crop_ratio`` then resize to ``resize_shape`` (reference :174-188).

        On the device, where no gradient with respect to the image is wanted (inference, and training on images that do not
        require grad), crop + antialiased resize run as one kernel (``eqa_crop_resize_aa``)."""
        crop, resize = self.crop_canonization, self.resize_canonization
        if (x.is_cuda and x.dtype == torch.float32 and not (torch.is_grad_enabled() and x.requires_grad) and isinstance(resize, Resize)
                and resize.antialias):
            from equiadapt_amd.images.transforms import resized_output_size

            out_hw = resized_output_size(crop.size, resize.size)
            if tuple(out_hw) == tuple(x.shape[-2:]) == tuple(crop.size):
                # crop ratio 1 and a resize to the size the image already has (the CIFAR-shaped configuration): CenterCrop returns
                # the image and torchvision's resize returns its input when the size matches -- no kernel at all
                return x
            # down-sampling (the reference's configurations) and up-sampling alike (the reference tutorial crops 64 -> 58 and
            # resizes back to 64: torch's antialiased filter has support 1 there, three taps): the tables follow torch's formula
            # for either direction (tests/test_abi_and_host.py, tests/test_gpu_parity.py::test_crop_resize_aa_matches_torch_interpolate)
            key = (tuple(x.shape[-2:]), crop.size, out_hw, str(x.device))
            tabs = self._consts.get(key)
            if tabs is None:
                t = geometry.aa_resize_tables(tuple(x.shape[-2:]), crop.size, out_hw)
                tabs = tuple(v.to(x.device) if isinstance(v, torch.Tensor) else v for v in t)
                self._consts[key] = tabs
            return ops.crop_resize_aa(x, tabs, out_hw)
        return resize(crop(x))

    # -- the hot path ------------------------------------------------------------------------------

    def canonicalize(self, x: torch.Tensor, targets: Optional[List] = None, **kwargs: Any
                     ) -> Union[torch.Tensor, Tuple[torch.Tensor, List]]:
        """Rotate (and reflect) every image into its canonical orientation (reference :190-238)."""
        self.device = x.device
        element = self.get_groupelement(x)
        gidx = self.canonicalization_info_dict["group_index"]
        reflections = "reflection" in element
        H, W = x.shape[-2:]
        pad = self.pad_size
        theta, flags = device_tables("canonicalize", self.num_rotations, reflections, (H + 2 * pad, W + 2 * pad), x.device)
        x = _CanonTransformFn.apply(x, element["rotation"], element.get("reflection"), gidx, theta, flags, pad,
                                    self.num_rotations)

        if targets:
            # boxes and masks follow the image (reference :217-236).  NOTE (reference behaviour kept): when the
            # group has reflections every target is flipped, whatever its own reflection indicator says.
            image_width = x.shape[-1]
            box_list = [t_["boxes"] for t_ in targets]
            counts = [int(b.shape[0]) for b in box_list]
            if sum(counts) > 0 and all(b.is_cuda and b.dim() == 2 and b.shape[1] == 4 and b.dtype == torch.float32 for b in box_list) \
                    and element["rotation"].dtype == torch.float32:
                # every box of the batch in one launch (the reference loops over samples, a dozen launches each); the same
                # fp32 arithmetic in the same order
                for t_, nb in zip(targets, canonicalize_boxes(box_list, element["rotation"].detach(), image_width, reflections)):
                    t_["boxes"] = nb
            else:
                if reflections:
                    for t in range(len(targets)):
                        targets[t]["boxes"] = flip_boxes(targets[t]["boxes"], image_width)
                for t in range(len(targets)):
                    targets[t]["boxes"] = rotate_boxes(targets[t]["boxes"], element["rotation"][t], image_width)
            masks = [t_["masks"] for t_ in targets]
            if all(m.is_cuda and m.dtype == torch.uint8 and m.dim() == 3 and m.shape[-2:] == masks[0].shape[-2:] for m in masks):
                # every mask of the batch in one nearest-neighbour kernel launch, element index read on the device
                new = canonicalize_masks(masks, gidx, self.num_rotations, flip_all=reflections)
                for t in range(len(targets)):
                    targets[t]["masks"] = new[t]
            else:  # other dtypes: op-by-op, one device->host copy for the whole batch
                rot_host = element["rotation"].detach().cpu()
                for t in range(len(targets)):
                    m = flip_masks(targets[t]["masks"]) if reflections else targets[t]["masks"]
                    targets[t]["masks"] = rotate_masks(m, -rot_host[t].item())
            return x, targets
        return x

    def invert_canonicalization(self, x_canonicalized_out: torch.Tensor, **kwargs: Any) -> torch.Tensor:
        """Map an image-shaped output back to the input's orientation (reference :240-259)."""
        induced_rep_type = kwargs.get("induced_rep_type", "regular")
        element = dict(self.canonicalization_info_dict["group_element"])
        element["group_index"] = self.canonicalization_info_dict["group_index"]
        return get_action_on_image_features(
            feature_map=x_canonicalized_out,
            group_info_dict=self.group_info_dict,
            group_element_dict=element,
            induced_rep_type=induced_rep_type,
        )


def _as_size(resize_shape: Any):
    if isinstance(resize_shape, int):
        return resize_shape
    return tuple(int(s) for s in resize_shape)


def _num_group(group_type: str, num_rotations: int) -> int:
    return num_rotations if group_type == "rotation" else 2 * num_rotations


class GroupEquivariantImageCanonicalization(DiscreteGroupImageCanonicalization):
    """Canonicalizer driven by a group-EQUIVARIANT network that outputs (B, G) activations.

    Reference: discrete_group.py:262-317.  ``group_type`` / ``num_rotations`` are read from the network.
    """

    def __init__(self, canonicalization_network: torch.nn.Module, canonicalization_hyperparams: Any, in_shape: tuple):
        super().__init__(canonicalization_network, canonicalization_hyperparams, in_shape)
        self.group_type = canonicalization_network.group_type
        self.num_rotations = canonicalization_network.num_rotations
        self.num_group = _num_group(self.group_type, self.num_rotations)
        self.group_info_dict = {"num_rotations": self.num_rotations, "num_group": self.num_group}

    def get_group_activations(self, x: torch.Tensor) -> torch.Tensor:
        x = self.transformations_before_canonicalization_network_forward(x)
        return self.canonicalization_network(x)


class OptimizedGroupEquivariantImageCanonicalization(DiscreteGroupImageCanonicalization):
    """Canonicalizer for a NON-equivariant network: score every group view against a reference vector.

    Reference: discrete_group.py:320-512.  The orbit (all G views of the resized batch) is written by one
    kernel launch (``eqa_orbit_expand_fwd``) instead of G x (pad, rotate, flip, crop) + ``torch.cat``.
    """

    def __init__(self, canonicalization_network: torch.nn.Module, canonicalization_hyperparams: Any, in_shape: tuple):
        super().__init__(canonicalization_network, canonicalization_hyperparams, in_shape)
        self.group_type = _get(canonicalization_hyperparams, "group_type")
        self.num_rotations = _get(canonicalization_hyperparams, "num_rotations")
        self.artifact_err_wt = _get(canonicalization_hyperparams, "artifact_err_wt")
        self.num_group = _num_group(self.group_type, self.num_rotations)
        self.out_vector_size = canonicalization_network.out_vector_size
        size = _get(canonicalization_hyperparams, "resize_shape")
        self.group_augment_size = int(size)
        gray = self.in_shape[0] == 1
        self.group_augment_pad = 0 if gray else math.ceil(self.group_augment_size * 0.5)
        self.crop_group_augment = torch.nn.Identity() if gray else CenterCrop(self.group_augment_size)
        self.pad_group_augment = torch.nn.Identity() if gray else EdgePad(self.group_augment_pad)
        self.reference_vector = torch.nn.Parameter(
            torch.randn(1, self.out_vector_size), requires_grad=_get(canonicalization_hyperparams, "learn_ref_vec"))
        self.group_info_dict = {"num_rotations": self.num_rotations, "num_group": self.num_group}

    def group_augment(self, x: torch.Tensor) -> torch.Tensor:
        """(B, C, s, s) -> (G*B, C, s, s), element-major (reference :387-427)."""
        s = x.shape[-1]
        pad = self.group_augment_pad
        theta, flags = device_tables("orbit", self.num_rotations, self.group_type == "roto-reflection",
                                     (s + 2 * pad, s + 2 * pad), x.device)
        return ops.orbit_expand(x, theta, flags, pad)

    def _rotate_batch(self, x: torch.Tensor, rot_index: torch.Tensor, sign: float) -> torch.Tensor:
        """pad -> rotate(sign * index * 360/N) -> crop for a per-image rotation index (artifact branch)."""
        s = x.shape[-1]
        pad = self.group_augment_pad
        theta = geometry.rotation_theta(sign * geometry.group_angles(self.num_rotations) * 1.0, (s + 2 * pad, s + 2 * pad))
        return ops.group_action(x, rot_index.to(torch.int32), theta.to(x.device), None, None, pad, (s, s), (pad, pad))

    def get_group_activations(self, x: torch.Tensor) -> torch.Tensor:
        """Cosine similarity of every view's embedding with the reference vector -> (B, G) (:429-481)."""
        x = self.transformations_before_canonicalization_network_forward(x)
        x_augmented = self.group_augment(x)
        vector_out = self.canonicalization_network(x_augmented)
        self.canonicalization_info_dict = {"vector_out": vector_out}

        if self.artifact_err_wt:
            # a random rotation and back, to penalise interpolation artifacts (reference :448-473)
            rot_idx = torch.randint(0, self.num_rotations, (x_augmented.shape[0],), device=x.device)
            x_dummy = self._rotate_batch(x_augmented, rot_idx, -1.0)
            x_dummy = self._rotate_batch(x_dummy, rot_idx, +1.0)
            vector_out_dummy = self.canonicalization_network(x_dummy)
            self.canonicalization_info_dict.update({"vector_out_dummy": vector_out_dummy})

        if vector_out.is_cuda and vector_out.dtype == torch.float32 and not torch.is_grad_enabled():
            # inference: cosine similarity + (G, B) -> (B, G) in one launch (the element-wise form below is ~10)
            return ops.cosine_group_activations(vector_out, self.reference_vector.detach(), self.num_group)
        scalar_out = F.cosine_similarity(self.reference_vector.repeat(vector_out.shape[0], 1), vector_out)
        return scalar_out.reshape(self.num_group, -1).T

    def get_optimization_specific_loss(self) -> torch.Tensor:
        """mean |off-diagonal Gram of the G view embeddings| + artifact_err_wt * MSE (reference :483-512)."""
        vectors = self.canonicalization_info_dict["vector_out"]
        artifact = 0
        if self.artifact_err_wt:
            artifact = F.mse_loss(self.canonicalization_info_dict["vector_out_dummy"], vectors)
        v = vectors.reshape(self.num_group, -1, self.out_vector_size).permute((1, 0, 2))
        gram = v @ v.permute((0, 2, 1))
        mask = 1.0 - torch.eye(self.num_group, device=self.device)
        return torch.abs(gram * mask).mean() + self.artifact_err_wt * artifact
