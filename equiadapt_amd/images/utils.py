"""Group actions on image-shaped outputs and on detection targets.

Reference: equiadapt/images/utils.py.  ``get_action_on_image_features`` is the body of
``invert_canonicalization``; it runs as ONE fused HIP kernel (rotate + flip + regular-representation
roll in a single gather) instead of grid_sample + blend + an int64 index tensor + gather.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from equiadapt_amd import ops
from equiadapt_amd.images import geometry

_device_tables: Dict[tuple, tuple] = {}


def device_tables(kind: str, num_rotations: int, reflections: bool, frame_hw: Tuple[int, int], device: torch.device):
    """Group-element tables for ``kind`` in {"canonicalize", "invert", "orbit"}, cached per device."""
    key = (kind, num_rotations, reflections, tuple(frame_hw), str(device))
    hit = _device_tables.get(key)
    if hit is None:
        build = {"canonicalize": geometry.canonicalize_tables, "invert": geometry.invert_tables,
                 "orbit": geometry.orbit_tables}[kind]
        hit = tuple(t.to(device) for t in build(num_rotations, reflections, tuple(frame_hw)))
        if num_rotations in (1, 2, 4) and frame_hw[0] == frame_hw[1]:
            # every element is a multiple of 90 degrees and the frame is square: a 32 x 32 output tile samples a window of at most
            # 35 x 35 source pixels (eqa_group_action_fwd_hint then reserves less LDS per block: more blocks per CU).  On a
            # non-square frame torch's normalised grid stretches a quarter turn by the aspect ratio -- a tile's window grows to
            # 36-47 rows, every such tile would fall to the direct gather path -- so no hint there.
            ops.register_window_hint(hit[0], 35)
        _device_tables[key] = hit
    return hit


def group_index_from_element(group_element_dict: dict, num_rotations: int) -> torch.Tensor:
    """Recover the int32 element index from {"rotation" (deg)[, "reflection"]} (no host sync).

    In training mode the reference's straight-through sums are 1 ulp off the table values; rounding to
    the nearest element makes the index exact again.
    """
    if "group_index" in group_element_dict:
        return group_element_dict["group_index"]
    rot = group_element_dict["rotation"].detach()
    idx = torch.round(rot / 360.0 * num_rotations).to(torch.int32) % num_rotations
    if "reflection" in group_element_dict:
        idx = idx + num_rotations * torch.round(group_element_dict["reflection"].detach()).to(torch.int32)
    return idx.to(torch.int32)


class _InvertActionFn(torch.autograd.Function):
    """out = roll(flip?(rotate(f, +rotation))).  Backward: adjoint scatter for d/d f (the prediction network's output
    usually needs it), d/d rotation, and d/d reflection = <g, out(r=1) - out(r=0)> (see _CanonTransformFn)."""

    @staticmethod
    def forward(ctx, feature_map, rotation, reflection, gidx, theta, flags, chan_map, num_rotations):
        none = torch.empty(0)
        ctx.save_for_backward(feature_map, gidx, theta, flags if flags is not None else none,
                              chan_map if chan_map is not None else none)
        ctx.has = (flags is not None, chan_map is not None)
        ctx.num_rotations = num_rotations
        return ops.invert_action(feature_map, gidx, theta, flags, chan_map)

    @staticmethod
    def backward(ctx, grad_out):
        f, gidx, theta, flags, chan_map = ctx.saved_tensors
        flags = flags if ctx.has[0] else None
        chan_map = chan_map if ctx.has[1] else None
        N = ctx.num_rotations
        need_f, need_rot, need_ref = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        grad_out = grad_out.contiguous()
        gf = g_rot = g_ref = None
        if need_f or need_rot:
            gf, g_rot = ops.group_action_bwd(f, grad_out, gidx, theta, flags, chan_map, 0, (0, 0), need_f, need_rot)
        if need_ref:
            ridx = gidx % N
            o0 = ops.invert_action(f, ridx, theta, flags, chan_map)
            o1 = ops.invert_action(f, ridx + N, theta, flags, chan_map)
            g_ref = (grad_out * (o1 - o0)).sum(dim=(1, 2, 3))
        return gf, g_rot, g_ref, None, None, None, None, None


def get_action_on_image_features(feature_map: torch.Tensor, group_info_dict: dict, group_element_dict: dict,
                                 induced_rep_type: str = "regular") -> torch.Tensor:
    """Apply the group element to a feature map (reference: images/utils.py:32-94).

    "scalar": rotate(+angle) with zero corners, then horizontal flip for elements whose reflection
    indicator is 0 (reference convention).  "regular": the same plus a cyclic roll of the group axis of
    ``(B, C/G, G, H, W)``.  "vector": NotImplementedError, anything else: ValueError -- as the reference.
    """
    num_rotations = group_info_dict["num_rotations"]
    num_group = group_info_dict["num_group"]
    assert len(feature_map.shape) == 4
    if induced_rep_type == "vector":
        raise NotImplementedError("Action for vector representation is not implemented")
    if induced_rep_type not in ("regular", "scalar"):
        raise ValueError("induced_rep_type must be regular, scalar or vector")
    C, H, W = feature_map.shape[1:]
    reflections = "reflection" in group_element_dict
    tables = device_tables("invert", num_rotations, reflections, (H, W), feature_map.device)
    theta, flags, chan_map = tables
    if induced_rep_type == "regular":
        assert C % num_group == 0
    else:
        chan_map = None
    gidx = group_index_from_element(group_element_dict, num_rotations)
    return _InvertActionFn.apply(feature_map, group_element_dict.get("rotation"), group_element_dict.get("reflection"),
                                 gidx, theta, flags, chan_map, num_rotations)


def roll_by_gather(feature_map: torch.Tensor, shifts: torch.Tensor) -> torch.Tensor:
    """out[:, :, g] = in[:, :, (g - shifts.long()) mod G] on (B, C, G, H, W) (reference: images/utils.py:8-29).

    Stand-alone form for API parity (the hot path fuses the roll into the invert kernel).  No
    (B,C,G,H,W) int64 index tensor is materialised: the permutation is applied on the group axis only.
    """
    B, C, G, H, W = feature_map.shape
    src = (torch.arange(G, device=feature_map.device)[None, :] - shifts[:, None].long()) % G  # (B, G)
    return torch.gather(feature_map, 2, src[:, None, :, None, None].expand(B, C, G, H, W))


def flip_boxes(boxes: torch.Tensor, width: int) -> torch.Tensor:
    """Horizontal flip of (n, 4) xyxy boxes, IN PLACE like the reference (images/utils.py:97-109)."""
    boxes[:, [0, 2]] = width - boxes[:, [2, 0]]
    return boxes


def flip_masks(masks: torch.Tensor) -> torch.Tensor:
    """images/utils.py:112-122."""
    return masks.flip(-1)


def rotate_masks(masks: torch.Tensor, angle: float) -> torch.Tensor:
    """Nearest-neighbour rotation of (n, H, W) masks about the image centre, zero fill.

    Semantics of torchvision.transforms.functional.rotate defaults (reference: images/utils.py:125-136):
    half-pixel grid, align_corners=False, round-half-even nearest, result rounded and cast back.
    """
    import math

    if masks.is_cuda and masks.dtype == torch.uint8 and masks.dim() == 3:
        rtheta = geometry.mask_rotation_table([float(angle)], tuple(masks.shape[-2:])).to(masks.device)
        eidx = torch.zeros(masks.shape[0], dtype=torch.int32, device=masks.device)
        return ops.mask_action_nearest(masks, eidx, rtheta, None)
    squeeze = masks.dim() == 3
    img = masks.unsqueeze(0) if squeeze else masks
    out_dtype = img.dtype
    cast = not torch.is_floating_point(img)
    if cast:
        img = img.to(torch.float32)
    h, w = img.shape[-2:]
    rot = math.radians(-float(angle))
    theta = torch.tensor([math.cos(rot), math.sin(rot), 0.0, -math.sin(rot), math.cos(rot), 0.0],
                         dtype=img.dtype, device=img.device).reshape(1, 2, 3)
    base = torch.empty(1, h, w, 3, dtype=img.dtype, device=img.device)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + 0.5, w * 0.5 - 0.5, steps=w, device=img.device))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + 0.5, h * 0.5 - 0.5, steps=h, device=img.device).unsqueeze_(-1))
    base[..., 2].fill_(1)
    scale = torch.tensor([0.5 * w, 0.5 * h], dtype=img.dtype, device=img.device)
    grid = base.view(1, h * w, 3).bmm(theta.transpose(1, 2) / scale).view(1, h, w, 2).expand(img.shape[0], h, w, 2)
    out = F.grid_sample(img, grid, mode="nearest", padding_mode="zeros", align_corners=False)
    if cast:
        out = torch.round(out).to(out_dtype)
    return out.squeeze(0) if squeeze else out


def owner_table(counts, device) -> torch.Tensor:
    """int32 (sum(counts),): index of the sample each box / mask of a batch belongs to.  Built on the host from the list
    lengths (known without a sync) and cached per length pattern: a training loop sees the same few patterns again and again,
    and ``repeat_interleave`` with a device count tensor costs an upload plus three launches every call."""
    key = ("owner", tuple(counts), str(device))
    hit = _device_tables.get(key)
    if hit is None:
        if sum(1 for k in _device_tables if k[0] == "owner") > 256:
            for k in [k for k in _device_tables if k[0] == "owner"]:
                del _device_tables[k]
        hit = torch.repeat_interleave(torch.arange(len(counts), dtype=torch.int32), torch.tensor(counts)).to(device)
        _device_tables[key] = hit
    return hit


def canonicalize_boxes(box_list, rotation_deg: torch.Tensor, width: int, flip_all: bool):
    """All samples' (n_t, 4) boxes in ONE launch (``eqa_boxes_action``): flipped when the group has reflections (every box,
    the reference's behaviour), then rotated by the sample's angle.  The caller's tensors receive the flipped values in place,
    as in the reference.  Reference: discrete_group.py:217-236 with flip_boxes / rotate_boxes (images/utils.py:97-109,161-187)."""
    counts = [int(b.shape[0]) for b in box_list]
    all_boxes = torch.cat(list(box_list), dim=0).contiguous()
    new, flipped = ops.boxes_action(all_boxes, owner_table(counts, all_boxes.device), rotation_deg.contiguous(), width, flip_all)
    if flipped is not None:
        torch._foreach_copy_(list(box_list), list(flipped.split(counts)))
    return list(new.split(counts))


def canonicalize_masks(mask_list, group_index: torch.Tensor, num_rotations: int, flip_all: bool):
    """All samples' masks in ONE launch: mask k of sample t is (flipped if ``flip_all``, the reference's behaviour
    whenever the group has reflections, then) rotated by -angle of sample t's group element.  No host sync: the element
    index stays on the device.  Reference: discrete_group.py:217-236 (per-sample Python loop with ``.item()``)."""
    counts = [int(m.shape[0]) for m in mask_list]
    if sum(counts) == 0:
        return list(mask_list)
    H, W = mask_list[0].shape[-2:]
    dev = mask_list[0].device
    key = ("mask", num_rotations, flip_all, (H, W), str(dev))
    hit = _device_tables.get(key)
    if hit is None:
        ang = geometry.group_angles(num_rotations)
        rtheta = geometry.mask_rotation_table((-ang).tolist(), (H, W)).to(dev)
        flags = torch.full((num_rotations,), geometry.FLIP_SRC if flip_all else 0, dtype=torch.int32, device=dev)
        hit = (rtheta, flags)
        _device_tables[key] = hit
    rtheta, flags = hit
    ridx = (group_index % num_rotations).to(torch.int32)
    eidx = ridx[owner_table(counts, dev).long()]
    if W % 16 == 0 and len(mask_list) > 1 and all(m.data_ptr() % 4 == 0 for m in mask_list):
        out = ops.mask_action_nearest_planes([m for m in mask_list if m.shape[0]], eidx, rtheta, flags)   # no concatenation pass
    else:
        out = ops.mask_action_nearest(torch.cat(list(mask_list), dim=0).contiguous(), eidx, rtheta, flags)
    return list(torch.split(out, counts, dim=0))


def rotate_points(origin: List[float], point: torch.Tensor, angle: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rotate (px, py) about ``origin`` by ``angle`` radians (images/utils.py:139-158)."""
    ox, oy = origin
    px, py = point
    c, s = torch.cos(angle), torch.sin(angle)
    return ox + c * (px - ox) - s * (py - oy), oy + s * (px - ox) + c * (py - oy)


def rotate_boxes(boxes: torch.Tensor, angle: torch.Tensor, width: int) -> torch.Tensor:
    """Rotate the two box corners about (W/2, W/2) by ``angle`` degrees and re-sort (images/utils.py:161-187)."""
    origin = [width / 2, width / 2]
    rad = torch.deg2rad(angle)
    x0, y0 = rotate_points(origin, boxes[:, :2].T, rad)
    x1, y1 = rotate_points(origin, boxes[:, 2:].T, rad)
    return torch.stack([torch.min(x0, x1), torch.min(y0, y1), torch.max(x0, x1), torch.max(y0, y1)], dim=-1)
