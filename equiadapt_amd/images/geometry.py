"""Host logic: the per-group-element tables the resampling kernels consume (see include/eqa_hip.h).

For a discrete group there are only E distinct geometric maps, so everything the reference recomputes
per call and per sample (rotation matrix -> homography -> normalise -> inverse -> affine grid) is
evaluated ONCE per (group, frame size) here, in fp32 and with the same torch op chain kornia 0.7.0 uses
(``rotate`` -> ``get_rotation_matrix2d`` -> ``warp_affine`` -> ``normalize_homography`` -> ``inverse``),
so the E 2x3 matrices are bit-identical to what the reference would hand to ``F.affine_grid``.
The tables are E*6 floats / E ints / E*G ints; they live on the device as non-persistent module buffers.
"""
import functools
import os
import math
from typing import Optional, Tuple

import torch

FLIP_SRC = 1  # sample the horizontally flipped frame (flip applied before the rotation)
FLIP_DST = 2  # flip the result horizontally (flip applied after the rotation)


def group_angles(num_rotations: int) -> torch.Tensor:
    """Angles of the rotation subgroup in degrees: linspace(0, 360, N + 1)[:N].

    Reference: images/canonicalization/discrete_group.py:110-112.
    """
    return torch.linspace(0.0, 360.0, num_rotations + 1)[:num_rotations]


def _pixel_to_norm(height: int, width: int, eps: float = 1e-14) -> torch.Tensor:
    m = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]])
    m[0, 0] = m[0, 0] * 2.0 / (eps if width == 1 else width - 1.0)
    m[1, 1] = m[1, 1] * 2.0 / (eps if height == 1 else height - 1.0)
    return m.unsqueeze(0)


def rotation_theta(angles_deg: torch.Tensor, frame_hw: Tuple[int, int]) -> torch.Tensor:
    """(E,) angles -> (E, 6) normalised sampling matrices for ``rotate(img, angle)`` on an (Hp, Wp) frame.

    Mirrors kornia.geometry.transform.rotate as called at images/canonicalization/discrete_group.py:213
    and images/utils.py:57,82 (centre ((Wp-1)/2, (Hp-1)/2), bilinear, zeros, align_corners=True).
    """
    Hp, Wp = frame_hw
    ang = torch.deg2rad(angles_deg.to(torch.float32).reshape(-1))
    E = ang.shape[0]
    cos_a, sin_a = torch.cos(ang), torch.sin(ang)
    rot = torch.stack([cos_a, sin_a, -sin_a, cos_a], dim=-1).view(E, 2, 2)
    center = torch.tensor([float(Wp - 1) / 2, float(Hp - 1) / 2]).expand(E, -1)
    unit = (torch.zeros(E, 2, 2) + torch.eye(2)) * torch.ones_like(center).unsqueeze(2).repeat(1, 1, 2)
    sr = rot @ unit
    alpha, beta = sr[:, 0, 0], sr[:, 0, 1]
    cx, cy = center[..., 0], center[..., 1]
    one = torch.tensor(1.0)
    M = torch.zeros(E, 3, 3)
    M[:, 0:2, 0:2] = sr
    M[:, 0, 2] = (one - alpha) * cx - beta * cy
    M[:, 1, 2] = beta * cx + (one - alpha) * cy
    M[:, 2, 2] = 1.0
    norm = _pixel_to_norm(Hp, Wp)
    dst_norm_from_src_norm = norm @ (M @ torch.linalg.inv(norm))
    return torch.linalg.inv(dst_norm_from_src_norm)[:, :2, :].reshape(E, 6).contiguous()


@functools.lru_cache(maxsize=64)
def canonicalize_tables(num_rotations: int, reflections: bool, frame_hw: Tuple[int, int]):
    """I5 tables: element e = (reflection, rotation index); rotate by -angle, flip the SOURCE when reflected."""
    ang = group_angles(num_rotations)
    theta = rotation_theta(-ang, frame_hw)
    if reflections:
        theta = torch.cat([theta, theta], dim=0)
        flags = torch.cat([torch.zeros(num_rotations), torch.full((num_rotations,), FLIP_SRC)]).to(torch.int32)
    else:
        flags = torch.zeros(num_rotations, dtype=torch.int32)
    return theta, flags


@functools.lru_cache(maxsize=64)
def orbit_tables(num_rotations: int, reflections: bool, frame_hw: Tuple[int, int]):
    """I8 tables: rotate by -angle, flip the RESULT for the reflected half (discrete_group.py:400-408)."""
    ang = torch.linspace(0, 360, num_rotations + 1)[:-1]
    theta = rotation_theta(-ang, frame_hw)
    if reflections:
        theta = torch.cat([theta, theta], dim=0)
        flags = torch.cat([torch.zeros(num_rotations), torch.full((num_rotations,), FLIP_DST)]).to(torch.int32)
    else:
        flags = torch.zeros(num_rotations, dtype=torch.int32)
    return theta, flags


@functools.lru_cache(maxsize=64)
def invert_tables(num_rotations: int, reflections: bool, frame_hw: Tuple[int, int]):
    """I7 tables (images/utils.py:54-89).

    rotate by +angle; when the group has reflections the result is flipped for elements whose reflection
    indicator is ZERO (``x*r + hflip(x)*(1-r)``) -- the reference's convention, opposite to I5.
    chan_map[e, g] = input group slot feeding output slot g for "regular" features:
    ``shift = angles/360*N``; first half rolled by ``shift.long()``, reflected half by ``(-shift).long()``
    (roll_by_gather, images/utils.py:8-29: out[g] = in[(g - s) mod N]).
    """
    N = num_rotations
    ang = group_angles(N)
    theta = rotation_theta(ang, frame_hw)
    shift = ang / 360.0 * N
    s_pos, s_neg = shift.long(), (-shift).long()
    g = torch.arange(N)
    first = (g[None, :] - s_pos[:, None]) % N          # (N elements, N slots)
    if not reflections:
        return theta, torch.zeros(N, dtype=torch.int32), first.to(torch.int32)
    second = N + (g[None, :] - s_neg[:, None]) % N
    per_rot = torch.cat([first, second], dim=1)       # (N, 2N)
    chan_map = torch.cat([per_rot, per_rot], dim=0)    # elements e and e+N share the rotation index
    theta = torch.cat([theta, theta], dim=0)
    flags = torch.cat([torch.full((N,), FLIP_DST), torch.zeros(N)]).to(torch.int32)
    return theta, flags, chan_map.to(torch.int32).contiguous()


def center_crop_offset(full: int, crop: int) -> int:
    """torchvision CenterCrop: int(round((full - crop) / 2.0)) with Python's round-half-even."""
    return int(round((full - crop) / 2.0))


# ---- I1: antialiased bilinear resize taps (torch aten/native/cpu/UpSampleKernel.cpp, HelperInterpBase) -------------


def _aa_axis(in_size: int, out_size: int, offset: int):
    """Per output index: first input tap (+ crop ``offset``) and K normalised triangle weights, as torch computes them
    for ``F.interpolate(mode="bilinear", antialias=True, align_corners=False)`` on fp32 data."""
    import numpy as np

    f32 = np.float32
    scale = f32(in_size) / f32(out_size)                      # area_pixel_compute_scale, no explicit scale factor
    support = f32(1.0) * scale if scale >= 1.0 else f32(1.0)  # interp_size * 0.5 = 1 for bilinear
    K = int(math.ceil(float(support))) * 2 + 1
    invscale = f32(1.0) / scale if scale >= 1.0 else f32(1.0)
    w = np.zeros((out_size, K), dtype=np.float32)
    start = np.zeros((out_size,), dtype=np.int32)
    for i in range(out_size):
        center = scale * f32(i + 0.5)
        xmin = max(int(float(center - support) + 0.5), 0)
        xsize = min(int(float(center + support) + 0.5), in_size) - xmin
        xsize = min(max(xsize, 0), K)
        total = f32(0.0)
        for j in range(xsize):
            t = f32(abs((float(f32(j + xmin) - center) + 0.5) * float(invscale)))
            wj = f32(1.0) - t if t < 1.0 else f32(0.0)
            w[i, j] = wj
            total = f32(total + wj)
        if total != 0.0:
            w[i, :xsize] = w[i, :xsize] / total
        start[i] = xmin + offset
    return w, start, K


@functools.lru_cache(maxsize=32)
def aa_resize_tables(in_hw: Tuple[int, int], crop_hw: Tuple[int, int], out_hw: Tuple[int, int], band: int = 8):
    """Tables for ``eqa_crop_resize_aa``: centre crop (torchvision offset rule) folded into the tap starts."""
    H, W = in_hw
    ch, cw = crop_hw
    top, left = center_crop_offset(H, ch), center_crop_offset(W, cw)
    wy, y0, Ky = _aa_axis(ch, out_hw[0], top)
    wx, x0, Kx = _aa_axis(cw, out_hw[1], left)
    K = max(Kx, Ky)
    import numpy as np

    def pad(w):
        out = np.zeros((w.shape[0], K), dtype=np.float32)
        out[:, : w.shape[1]] = w
        return out

    max_rows = 1
    for r0 in range(0, out_hw[0], band):
        r1 = min(r0 + band, out_hw[0])
        max_rows = max(max_rows, min(int(y0[r1 - 1]) + K, H) - int(y0[r0]))
    # column range any output column reads (for the wide-filter kernel, which stages input rows in LDS)
    x_begin = int(x0.min())
    x_span = min(int(x0.max()) + K, W) - x_begin
    return (torch.from_numpy(pad(wx)), torch.from_numpy(x0), torch.from_numpy(pad(wy)), torch.from_numpy(y0), K, max_rows,
            x_begin, x_span)


# ---- I6: torchvision nearest-neighbour rotation of masks -------------------------------------------------------------


def mask_rotation_table(angles_deg, hw: Tuple[int, int]) -> torch.Tensor:
    """(E, 6) rescaled inverse affine matrices of ``transforms.functional.rotate(mask, angle)`` (nearest, no expand):
    _get_inverse_affine_matrix(center 0, -angle) then ``theta^T / (0.5 w, 0.5 h)`` -- order r00,r10,r20,r01,r11,r21."""
    h, w = hw
    rows = []
    for a in angles_deg:
        rot = math.radians(-float(a))
        c, s_ = math.cos(rot), math.sin(rot)
        rows.append([c, s_, 0.0, -s_, c, 0.0])
    theta = torch.tensor(rows, dtype=torch.float32).reshape(-1, 2, 3)
    resc = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32)   # (E, 3, 2)
    return torch.stack([resc[:, 0, 0], resc[:, 1, 0], resc[:, 2, 0], resc[:, 0, 1], resc[:, 1, 1], resc[:, 2, 1]], dim=1).contiguous()


def warp_affine_theta(M: torch.Tensor, frame_hw: Tuple[int, int]) -> torch.Tensor:
    """Per-sample pixel-space matrices of ``K.geometry.warp_affine(x, M, dsize=frame)`` -> (B, 6) rows for the kernels'
    ``affine_grid(align_corners=True)`` arithmetic.  Differentiable torch ops on M's device (fp64 inside).

    kornia 0.7.0 ``warp_affine``: ``dst_norm_trans_src_norm = N M N^-1`` (``normalize_homography``), inverted, first two
    rows to ``F.affine_grid``.  With ``M = [A | t]`` and ``N p = s * p - 1``, ``s = (2/(W-1), 2/(H-1))``:
    ``theta = [S A^-1 S^-1 | S A^-1 (1/s) - S A^-1 t - 1]`` in closed form (reference use: continuous_group.py:203).
    """
    H, W = frame_hw
    Md = M.double()
    a, b, c, d = Md[:, 0, 0], Md[:, 0, 1], Md[:, 1, 0], Md[:, 1, 1]
    tx, ty = Md[:, 0, 2], Md[:, 1, 2]
    det = a * d - b * c
    ia, ib, ic, id_ = d / det, -b / det, -c / det, a / det               # A^-1
    sx, sy = 2.0 / max(W - 1, 1e-14), 2.0 / max(H - 1, 1e-14)
    itx, ity = -(ia * tx + ib * ty), -(ic * tx + id_ * ty)               # -A^-1 t
    r0 = torch.stack([ia, ib * sx / sy, sx * (ia / sx + ib / sy + itx) - 1.0], dim=1)
    r1 = torch.stack([ic * sy / sx, id_, sy * (ic / sx + id_ / sy + ity) - 1.0], dim=1)
    return torch.cat([r0, r1], dim=1).to(M.dtype)


def affine_grid_theta_half_pixel(theta: torch.Tensor, frame_hw: Tuple[int, int]) -> torch.Tensor:
    """(B, 2, 3) matrices meant for ``F.affine_grid / F.grid_sample`` with ``align_corners=False`` -> (B, 6) rows that give the
    same sampling positions under the kernels' ``align_corners=True`` arithmetic:  x_n^F = x_n^T (W-1)/W and
    ix = (s W + W - 1)/2 versus (s^T + 1)(W-1)/2  =>  s^T = s W/(W-1)  (reference use: continuous_group.py:386-387)."""
    H, W = frame_hw
    t = theta.double()
    kx, ky = W / max(W - 1, 1e-14), H / max(H - 1, 1e-14)
    r0 = torch.stack([t[:, 0, 0], t[:, 0, 1] * (H - 1) / H * kx, t[:, 0, 2] * kx], dim=1)
    r1 = torch.stack([t[:, 1, 0] * (W - 1) / W * ky, t[:, 1, 1], t[:, 1, 2] * ky], dim=1)
    return torch.cat([r0, r1], dim=1).to(theta.dtype)
