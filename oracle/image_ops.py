"""Oracle for the image rows of SURVEY.md section 8a (I1, I3-I9).  TEST INFRASTRUCTURE ONLY.

Two layers:

1. ``kornia_*`` / ``tv_*``: restatements of the third-party functions the reference calls
   (kornia 0.7.0 ``geometry.rotate`` / ``hflip``; torchvision 0.17.0 ``Pad`` / ``CenterCrop`` /
   ``Resize`` / ``functional.rotate``) on the torch primitives those libraries delegate to.
   **Parity unpinned** -- see ``oracle/__init__.py``.
2. the reference's op sequences on top of them, in the reference's own order and with the same
   intermediate tensors (so the same functions double as the timed "reference CPU path").

All arithmetic is fp32 like the reference (it never changes dtype).
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# layer 1: kornia 0.7.0
# --------------------------------------------------------------------------------------


def kornia_normal_transform_pixel(height: int, width: int, eps: float = 1e-14) -> torch.Tensor:
    """kornia.geometry.conversions.normal_transform_pixel: pixel -> [-1, 1] (align_corners) 3x3."""
    tr = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]])
    width_denom = eps if width == 1 else width - 1.0
    height_denom = eps if height == 1 else height - 1.0
    tr[0, 0] = tr[0, 0] * 2.0 / width_denom
    tr[1, 1] = tr[1, 1] * 2.0 / height_denom
    return tr.unsqueeze(0)


def kornia_rotation_matrix2d(center: torch.Tensor, angle_deg: torch.Tensor) -> torch.Tensor:
    """kornia.geometry.transform.get_rotation_matrix2d with unit scale -> (B, 2, 3).

    ``angle_to_rotation_matrix`` gives [[cos, sin], [-sin, cos]] (positive angle = counter-clockwise
    on screen).  The translation column keeps ``center`` fixed.
    """
    ang = torch.deg2rad(angle_deg)
    cos_a, sin_a = torch.cos(ang), torch.sin(ang)
    rot = torch.stack([cos_a, sin_a, -sin_a, cos_a], dim=-1).view(-1, 2, 2)
    scale = torch.ones_like(center)
    scaling = torch.zeros(center.shape[0], 2, 2) + torch.eye(2)
    scaling = scaling * scale.unsqueeze(2).repeat(1, 1, 2)
    scaled = rot @ scaling
    alpha, beta = scaled[:, 0, 0], scaled[:, 0, 1]
    x, y = center[..., 0], center[..., 1]
    one = torch.tensor(1.0)
    M = torch.zeros(center.shape[0], 2, 3)
    M[..., 0:2, 0:2] = scaled
    M[..., 0, 2] = (one - alpha) * x - beta * y
    M[..., 1, 2] = beta * x + (one - alpha) * y
    return M


def kornia_affine_theta(M: torch.Tensor, src_hw: Tuple[int, int], dst_hw: Tuple[int, int]) -> torch.Tensor:
    """The normalised 2x3 matrix kornia.warp_affine hands to ``F.affine_grid``.

    warp_affine: M -> homography -> normalize_homography -> inverse -> [:, :2, :].
    """
    B = M.shape[0]
    bottom = torch.tensor([0.0, 0.0, 1.0]).view(1, 1, 3).expand(B, 1, 3)
    M3 = torch.cat([M, bottom], dim=1)
    src_norm_trans_src_pix = kornia_normal_transform_pixel(*src_hw)
    src_pix_trans_src_norm = torch.linalg.inv(src_norm_trans_src_pix)
    dst_norm_trans_dst_pix = kornia_normal_transform_pixel(*dst_hw)
    dst_norm_trans_src_norm = dst_norm_trans_dst_pix @ (M3 @ src_pix_trans_src_norm)
    src_norm_trans_dst_norm = torch.linalg.inv(dst_norm_trans_src_norm)
    return src_norm_trans_dst_norm[:, :2, :]


def kornia_warp_affine(src: torch.Tensor, M: torch.Tensor, dsize: Tuple[int, int]) -> torch.Tensor:
    """kornia.geometry.transform.warp_affine defaults: bilinear, zeros, align_corners=True."""
    B, C, H, W = src.shape
    theta = kornia_affine_theta(M, (H, W), dsize)
    grid = F.affine_grid(theta, [B, C, dsize[0], dsize[1]], align_corners=True)
    return F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def kornia_rotate(x: torch.Tensor, angle_deg: torch.Tensor) -> torch.Tensor:
    """kornia.geometry.transform.rotate(tensor, angle): about ((W-1)/2, (H-1)/2), same size out.

    Called by the reference at images/canonicalization/discrete_group.py:213,404,456,463 and
    images/utils.py:57,82 and custom_group_equivariant_layers.py:77,184,313,481.
    """
    B, _, H, W = x.shape
    angle = torch.as_tensor(angle_deg, dtype=x.dtype).reshape(-1).expand(B)
    center = torch.tensor([float(W - 1) / 2, float(H - 1) / 2], dtype=x.dtype).expand(B, -1)
    M = kornia_rotation_matrix2d(center, angle)
    return kornia_warp_affine(x, M, (H, W))


def kornia_hflip(x: torch.Tensor) -> torch.Tensor:
    """kornia.geometry.transform.hflip: reverse the last (width) axis."""
    return x.flip(-1)


# --------------------------------------------------------------------------------------
# layer 1: torchvision 0.17.0 tensor transforms
# --------------------------------------------------------------------------------------


def tv_pad_edge(x: torch.Tensor, p: int) -> torch.Tensor:
    """transforms.Pad(p, padding_mode="edge") on a tensor -> F.pad(replicate) on all four sides."""
    return F.pad(x, (p, p, p, p), mode="replicate")


def tv_center_crop(x: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """transforms.CenterCrop: top/left = int(round((H - h) / 2.0)) (Python banker's rounding)."""
    H, W = x.shape[-2:]
    h, w = size
    assert h <= H and w <= W, "oracle covers crop <= image only (the reference's own use)"
    top = int(round((H - h) / 2.0))
    left = int(round((W - w) / 2.0))
    return x[..., top : top + h, left : left + w]


def tv_resize_output_size(hw: Tuple[int, int], size: Union[int, Sequence[int]]) -> Tuple[int, int]:
    """torchvision _compute_resized_output_size: int -> shorter edge, pair -> exact."""
    h, w = hw
    if isinstance(size, int) or len(size) == 1:
        s = size if isinstance(size, int) else size[0]
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = s, int(s * long / short)
        return (new_long, new_short) if w <= h else (new_short, new_long)
    return int(size[0]), int(size[1])


def tv_resize(x: torch.Tensor, size: Union[int, Sequence[int]], antialias: bool = True) -> torch.Tensor:
    """transforms.Resize on a float tensor: bilinear, align_corners=False, antialias (0.17 default True)."""
    oh, ow = tv_resize_output_size(tuple(x.shape[-2:]), size)
    return F.interpolate(x, size=[oh, ow], mode="bilinear", align_corners=False, antialias=antialias)


def tv_rotate_nearest(masks: torch.Tensor, angle_deg: float) -> torch.Tensor:
    """torchvision.transforms.functional.rotate defaults (nearest, no expand, zero fill) on (n,H,W).

    _get_inverse_affine_matrix(center=0, -angle) -> _gen_affine_grid (half-pixel base grid,
    theta / (0.5w, 0.5h)) -> grid_sample(nearest, zeros, align_corners=False) -> round -> cast back.
    Used by images/utils.py:125-136 (rotate_masks).
    """
    squeeze = masks.dim() == 3
    img = masks.unsqueeze(0) if squeeze else masks
    out_dtype = img.dtype
    need_cast = not torch.is_floating_point(img)
    if need_cast:
        img = img.to(torch.float32)
    h, w = img.shape[-2:]
    rot = math.radians(-angle_deg)
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    matrix = [d, -b, 0.0, -c, a, 0.0]
    theta = torch.tensor(matrix, dtype=img.dtype).reshape(1, 2, 3)
    base = torch.empty(1, h, w, 3, dtype=img.dtype)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, steps=h).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=img.dtype)
    grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
    grid = grid.expand(img.shape[0], h, w, 2)
    out = F.grid_sample(img, grid, mode="nearest", padding_mode="zeros", align_corners=False)
    if need_cast:
        out = torch.round(out).to(out_dtype)
    return out.squeeze(0) if squeeze else out


# --------------------------------------------------------------------------------------
# layer 2: reference op sequences
# --------------------------------------------------------------------------------------


def group_angles(num_rotations: int) -> torch.Tensor:
    """linspace(0, 360, N+1)[:N] -- images/canonicalization/discrete_group.py:110-112."""
    return torch.linspace(0.0, 360.0, num_rotations + 1)[:num_rotations]


def pre_canonicalization_transform(
    x: torch.Tensor, in_shape: Tuple[int, int, int], input_crop_ratio: float, resize_shape, antialias: bool = True
) -> torch.Tensor:
    """I1: CenterCrop(ceil(H*r), ceil(W*r)) then Resize; identity for grayscale.

    images/canonicalization/discrete_group.py:174-188, transforms built :73-92.
    """
    if in_shape[0] == 1:
        return x
    ch = math.ceil(in_shape[-2] * input_crop_ratio)
    cw = math.ceil(in_shape[-1] * input_crop_ratio)
    x = tv_center_crop(x, (ch, cw))
    return tv_resize(x, resize_shape, antialias=antialias)


def canonicalize_images(
    x: torch.Tensor, rotation_deg: torch.Tensor, reflection: Optional[torch.Tensor], in_shape: Tuple[int, int, int]
) -> torch.Tensor:
    """I5: pad(edge, ceil(W/2)) -> [reflect blend] -> rotate(-rotation) -> CenterCrop(H, W).

    images/canonicalization/discrete_group.py:204-215 (pad/crop built :62-71; identity if C == 1).
    """
    gray = in_shape[0] == 1
    if not gray:
        x = tv_pad_edge(x, math.ceil(in_shape[-1] * 0.5))
    if reflection is not None:
        r = reflection[:, None, None, None]
        x = (1 - r) * x + r * kornia_hflip(x)
    x = kornia_rotate(x, -rotation_deg)
    if not gray:
        x = tv_center_crop(x, (in_shape[-2], in_shape[-1]))
    return x


def roll_by_gather(feature_map: torch.Tensor, shifts: torch.Tensor) -> torch.Tensor:
    """images/utils.py:8-29: out[..., g, :, :] = in[..., (g - shifts.long()) % G, :, :] via an index tensor."""
    batch, channel, group, xd, yd = feature_map.shape
    ar = torch.arange(group).view(1, 1, group, 1, 1).repeat(batch, channel, 1, xd, yd)
    idx = (ar - shifts[:, None, None, None, None].long()) % group
    return torch.gather(feature_map, 2, idx)


def invert_action(
    feature_map: torch.Tensor,
    rotation_deg: torch.Tensor,
    reflection: Optional[torch.Tensor],
    num_rotations: int,
    num_group: int,
    induced_rep_type: str = "regular",
) -> torch.Tensor:
    """I7: get_action_on_image_features, images/utils.py:32-94.

    rotate(+angles) with zero corners; flip blend whose indicator is the OPPOSITE of I5
    (``x*r + hflip(x)*(1-r)``: flipped when r == 0); for "regular" a cyclic roll of the group axis by
    ``(angles/360*num_rotations).long()`` (second half by the negated shift when reflections exist).
    """
    assert feature_map.dim() == 4
    B, C, H, W = feature_map.shape
    if induced_rep_type not in ("regular", "scalar", "vector"):
        raise ValueError("induced_rep_type must be regular, scalar or vector")
    if induced_rep_type == "vector":
        raise NotImplementedError("Action for vector representation is not implemented")
    if induced_rep_type == "regular":
        assert C % num_group == 0
    x_out = kornia_rotate(feature_map, rotation_deg)
    if reflection is not None:
        r = reflection[:, None, None, None]
        x_out = x_out * r + kornia_hflip(x_out) * (1 - r)
    if induced_rep_type == "scalar":
        return x_out
    x_out = x_out.reshape(B, C // num_group, num_group, H, W)
    shift = rotation_deg / 360.0 * num_rotations
    if reflection is not None:
        x_out = torch.cat(
            [
                roll_by_gather(x_out[:, :, :num_rotations], shift),
                roll_by_gather(x_out[:, :, num_rotations:], -shift),
            ],
            dim=2,
        )
    else:
        x_out = roll_by_gather(x_out, shift)
    return x_out.reshape(B, -1, H, W)


def group_pool(feature_map: torch.Tensor) -> torch.Tensor:
    """I3: mean over (channel, H', W') of a (B, C, G, H', W') map -> (B, G).

    images/canonicalization_networks/escnn_networks.py:106-115, custom_equivariant_networks.py:91.
    """
    return torch.mean(feature_map, dim=(1, 3, 4))


def onehot_from_activations(
    group_activations: torch.Tensor, num_group: int, beta: float, training: bool, gradient_trick: str = "straight_through"
) -> torch.Tensor:
    """I4: common/basecanonicalization.py:221-256 (argmax one-hot, softmax, straight-through)."""
    hard = F.one_hot(torch.argmax(group_activations, dim=-1), num_group).float()
    soft = F.softmax(beta * group_activations, dim=-1)
    if gradient_trick == "straight_through":
        return hard + soft - soft.detach() if training else hard
    if gradient_trick == "gumbel_softmax":
        return F.gumbel_softmax(group_activations, tau=1, hard=True)
    raise ValueError(f"Gradient trick {gradient_trick} not implemented")


def group_element_from_activations(
    group_activations: torch.Tensor, num_rotations: int, group_type: str, beta: float, training: bool
) -> Dict[str, torch.Tensor]:
    """I4: images/canonicalization/discrete_group.py:94-135 -> {"rotation"[, "reflection"]}."""
    num_group = num_rotations if group_type == "rotation" else 2 * num_rotations
    onehot = onehot_from_activations(group_activations, num_group, beta, training)
    angles = group_angles(num_rotations)
    rot_comp = torch.cat([angles, angles], dim=0) if group_type == "roto-reflection" else angles
    out = {"rotation": torch.sum(onehot * rot_comp, dim=-1)}
    if group_type == "roto-reflection":
        ident = torch.cat([torch.zeros(num_rotations), torch.ones(num_rotations)], dim=0)
        out["reflection"] = torch.sum(onehot * ident, dim=-1)
    return out


def prior_regularization_loss(group_activations: torch.Tensor) -> torch.Tensor:
    """I9: CrossEntropy(activations, class 0) -- common/basecanonicalization.py:290-301."""
    target = torch.zeros((group_activations.shape[0],), dtype=torch.long)
    return torch.nn.CrossEntropyLoss()(group_activations, target)


def identity_metric(group_activations: torch.Tensor) -> torch.Tensor:
    """I9: mean(argmax == 0) -- common/basecanonicalization.py:303-311."""
    return (group_activations.argmax(dim=-1) == 0).float().mean()


def orbit_expand(x: torch.Tensor, num_rotations: int, group_type: str, size: int, gray: bool = False) -> torch.Tensor:
    """I8: group_augment / rotate_and_maybe_reflect, images/canonicalization/discrete_group.py:387-427.

    For every group element: pad(edge, ceil(size/2)) -> rotate(-deg) -> [hflip AFTER the rotation]
    -> CenterCrop(size); concatenated element-major along dim 0 -> (G*B, C, size, size).
    """
    degrees = torch.linspace(0, 360, num_rotations + 1)[:-1]

    def one_sweep(reflect: bool) -> List[torch.Tensor]:
        outs = []
        for deg in degrees:
            xr = x if gray else tv_pad_edge(x, math.ceil(size * 0.5))
            xr = kornia_rotate(xr, -deg)
            if reflect:
                xr = kornia_hflip(xr)
            xr = xr if gray else tv_center_crop(xr, (size, size))
            outs.append(xr)
        return outs

    views = one_sweep(False)
    if group_type == "roto-reflection":
        views += one_sweep(True)
    return torch.cat(views, dim=0)


def optimized_group_activations(vector_out: torch.Tensor, reference_vector: torch.Tensor, num_group: int) -> torch.Tensor:
    """I8 tail: cosine similarity with the reference vector, (G*B,) -> (B, G).

    images/canonicalization/discrete_group.py:475-481.
    """
    scalar = F.cosine_similarity(reference_vector.repeat(vector_out.shape[0], 1), vector_out)
    return scalar.reshape(num_group, -1).T


def optimization_specific_loss(
    vector_out: torch.Tensor, num_group: int, out_vector_size: int, artifact_err_wt: float = 0.0,
    vector_out_dummy: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """I8 loss: mean |off-diagonal Gram| + artifact_err_wt * MSE -- discrete_group.py:483-512."""
    err = 0
    if artifact_err_wt:
        err = F.mse_loss(vector_out_dummy, vector_out)
    v = vector_out.reshape(num_group, -1, out_vector_size).permute((1, 0, 2))
    d = v @ v.permute((0, 2, 1))
    mask = 1.0 - torch.eye(num_group)
    return torch.abs(d * mask).mean() + artifact_err_wt * err


# ---- targets (I6) ---------------------------------------------------------------------


def flip_boxes(boxes: torch.Tensor, width: int) -> torch.Tensor:
    """images/utils.py:97-109 (in place, like the reference)."""
    boxes[:, [0, 2]] = width - boxes[:, [2, 0]]
    return boxes


def flip_masks(masks: torch.Tensor) -> torch.Tensor:
    """images/utils.py:112-122."""
    return masks.flip(-1)


def rotate_masks(masks: torch.Tensor, angle_deg: float) -> torch.Tensor:
    """images/utils.py:125-136."""
    return tv_rotate_nearest(masks, angle_deg)


def rotate_boxes(boxes: torch.Tensor, angle_deg: torch.Tensor, width: int) -> torch.Tensor:
    """images/utils.py:139-187: rotate the two corners about (W/2, W/2), re-sort min/max."""
    ang = torch.deg2rad(angle_deg)
    ox = oy = width / 2

    def rot(pt):
        px, py = pt
        qx = ox + torch.cos(ang) * (px - ox) - torch.sin(ang) * (py - oy)
        qy = oy + torch.sin(ang) * (px - ox) + torch.cos(ang) * (py - oy)
        return qx, qy

    x0, y0 = rot(boxes[:, :2].T)
    x1, y1 = rot(boxes[:, 2:].T)
    x0, x1 = torch.min(x0, x1), torch.max(x0, x1)
    y0, y1 = torch.min(y0, y1), torch.max(y0, y1)
    return torch.stack([x0, y0, x1, y1], dim=-1)


# --------------------------------------------------------------------------------------
# independent fp64 ground truth for the resampling (used to size the fp32 error budget)
# --------------------------------------------------------------------------------------


def rotate_exact_fp64(x: torch.Tensor, angle_deg: torch.Tensor, pad: int = 0, crop_hw: Optional[Tuple[int, int]] = None,
                      pre_hflip: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Analytic restatement in fp64, written in pixel space (no normalised grid, no matrix inverse):

    value(i, j) = bilinear sample of the edge-padded (by ``pad``), optionally h-flipped frame at
    c + R(angle)^T-style inverse rotation of ((j, i) - c), zero outside the padded frame; then the
    centre crop.  Independent in *formulation* from ``kornia_rotate`` (same conventions).
    """
    xd = x.double()
    B, C, H, W = xd.shape
    Hp, Wp = H + 2 * pad, W + 2 * pad
    ch, cw = crop_hw if crop_hw is not None else (Hp, Wp)
    top, left = int(round((Hp - ch) / 2.0)), int(round((Wp - cw) / 2.0))
    ang = torch.deg2rad(torch.as_tensor(angle_deg, dtype=torch.float64).reshape(-1).expand(B))
    cos, sin = torch.cos(ang)[:, None, None], torch.sin(ang)[:, None, None]
    cx, cy = (Wp - 1) / 2.0, (Hp - 1) / 2.0
    ii = torch.arange(ch, dtype=torch.float64)[None, :, None] + top - cy
    jj = torch.arange(cw, dtype=torch.float64)[None, None, :] + left - cx
    # dst = M src with M = [[cos, sin], [-sin, cos]]  =>  src = M^-1 dst = [[cos, -sin], [sin, cos]] dst
    sx = cos * jj - sin * ii + cx
    sy = sin * jj + cos * ii + cy
    x0, y0 = torch.floor(sx), torch.floor(sy)
    fx, fy = sx - x0, sy - y0
    out = torch.zeros(B, C, ch, cw, dtype=torch.float64)
    bidx = torch.arange(B)[:, None, None].expand(B, ch, cw)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            px, py = (x0 + dx).long(), (y0 + dy).long()
            inside = (px >= 0) & (px < Wp) & (py >= 0) & (py < Hp)
            if pre_hflip is not None:
                flipped = pre_hflip.bool()[:, None, None]
                px = torch.where(flipped, Wp - 1 - px, px)
            qx = (px - pad).clamp(0, W - 1)
            qy = (py - pad).clamp(0, H - 1)
            vals = xd[bidx, :, qy, qx].permute(0, 3, 1, 2)  # (B, C, ch, cw)
            out += (wy * wx * inside)[:, None] * vals
    return out


def group_inference_orbit(x: torch.Tensor, num_rotations: int, group_type: str) -> torch.Tensor:
    """Row (f).2: the test-time orbit of GroupInference, examples/images/classification/inference_utils.py:100-123.

    pad(edge, ceil(0.4 H)) -> [hflip] -> torchvision rotate(+deg) (NEAREST on tensors) -> CenterCrop(H, W);
    returns (E, B, C, H, W) in the reference's element order (rotations, then rotations of the reflection).
    """
    B, C, H, W = x.shape
    pad = math.ceil(H * 0.4)
    degrees = torch.linspace(0, 360, num_rotations + 1)[:-1]
    xp = tv_pad_edge(x, pad)
    outs = []
    for flip in ([False, True] if group_type == "roto-reflection" else [False]):
        src = xp.flip(-1) if flip else xp
        for d in degrees:
            outs.append(tv_center_crop(tv_rotate_nearest(src, d.item()), (H, W)))
    return torch.stack(outs, dim=0)


# --------------------------------------------------------------------------------------
# continuous-group (SO(2)) image canonicalization -- images/canonicalization/continuous_group.py
# --------------------------------------------------------------------------------------


def steerable_rotation_from_vector(vectors: torch.Tensor) -> torch.Tensor:
    """(B, 2) -> (B, 2, 2) with rows v1 = v/|v| and v2 = (-v1_y, v1_x)  -- continuous_group.py:246-261, :331-346."""
    v1 = vectors / torch.norm(vectors, dim=1, keepdim=True)
    v2 = torch.stack([-v1[:, 1], v1[:, 0]], dim=1)
    return torch.stack([v1, v2], dim=1)


def canonicalize_images_continuous(x: torch.Tensor, rotation_matrices: torch.Tensor, gray: bool = False):
    """ContinuousGroupImageCanonicalization.canonicalize for group_type "rotation" -- continuous_group.py:162-210.

    Returns (canonical images, the matrices as the reference leaves them: the off-diagonal entries negated IN PLACE at
    :178, i.e. the inverse rotations -- the info dict of the reference holds these after the call).
    Note the centre (Hp // 2, Wp // 2) with the HEIGHT in the x role (:192), not kornia.rotate's ((W-1)/2, (H-1)/2).
    """
    R = rotation_matrices.clone()
    R[:, [0, 1], [1, 0]] *= -1
    H, W = x.shape[-2:]
    xp = x if gray else tv_pad_edge(x, math.ceil(W * 0.5))
    alpha, beta = R[:, 0, 0], R[:, 0, 1]
    cx, cy = xp.shape[-2] // 2, xp.shape[-1] // 2
    affine_part = torch.stack([(1 - alpha) * cx - beta * cy, beta * cx + (1 - alpha) * cy], dim=1)
    M = torch.cat([R, affine_part.unsqueeze(-1)], dim=-1)
    out = kornia_warp_affine(xp, M, (xp.shape[-2], xp.shape[-1]))
    return (out if gray else tv_center_crop(out, (H, W))), R


def continuous_group_augment(x: torch.Tensor, angles: torch.Tensor, gray: bool = False):
    """OptimizedSteerableImageCanonicalization.group_augment for group_type "rotation" with the random angles given
    -- continuous_group.py:348-398.  Reference behaviour kept: ``torch.stack((cos, -sin, sin, cos)).reshape(-1, 2, 2)``
    (:365-367) reshapes a (4, B) tensor, so for B > 1 the "rotation matrices" mix the samples' sines and cosines; the
    returned ground-truth matrices are those mixed matrices with the off-diagonal negated (:394)."""
    B = x.shape[0]
    cos_a, sin_a = torch.cos(angles), torch.sin(angles)
    rot = torch.zeros(B, 2, 3, dtype=x.dtype)
    rot[:, :2, :2] = torch.stack((cos_a, -sin_a, sin_a, cos_a)).reshape(-1, 2, 2)
    H, W = x.shape[-2:]
    xp = x if gray else tv_pad_edge(x, math.ceil(W * 0.5))
    grid = F.affine_grid(rot, list(xp.size()), align_corners=False)
    aug = F.grid_sample(xp, grid, align_corners=False)
    aug = aug if gray else tv_center_crop(aug, (H, W))
    rot[:, [0, 1], [1, 0]] *= -1
    return aug, rot[:, :, :2]


def continuous_optimization_loss(rep_augmented: torch.Tensor, rep_augmented_gt: torch.Tensor) -> torch.Tensor:
    """OptimizedSteerableImageCanonicalization.get_optimization_specific_loss -- continuous_group.py:472-497."""
    return F.mse_loss(rep_augmented, rep_augmented_gt)
