"""GPU: size-independent properties of the hot path at BASELINE.json's FULL sizes (256 x 3 x 224 x 224 C8, 32 x 1024 x 1024 D4
masks, 2048 clouds of 1024 points), where the CPU oracle would take minutes: exact permutations for right angles, invert o
canonicalize = identity, linearity of the resampling, checksums, invariance of the canonical form under the group."""
import math
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

import bench  # noqa: E402  (build_canonicalizer: the headline configuration)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    from equiadapt_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def _tables(kind, N, refl, hw, dev):
    from equiadapt_amd.images.utils import device_tables

    return device_tables(kind, N, refl, hw, dev)


def test_right_angle_elements_are_exact_permutations_full_size(dev):
    """C8 elements 0, 90, 180, 270 degrees on the headline batch: pad -> rotate(-theta) -> crop is torch.rot90 (every sample point
    is a pixel centre up to the rounding of the reference's normalised-grid arithmetic, which the kernel reproduces: the stated
    white-noise tolerance 1e-3 max / 1e-4 rms, tests/test_gpu_parity.py), invert undoes it, and the checksum of every image is
    unchanged to 1e-5 of its pixel mass."""
    from equiadapt_amd import ops

    torch.manual_seed(100)
    B, C, H, W, N = 256, 3, 224, 224, 8
    x = torch.randn(B, C, H, W, device=dev)
    pad = math.ceil(W * 0.5)
    th, fl = _tables("canonicalize", N, False, (H + 2 * pad, W + 2 * pad), dev)
    thi, fli, _ = _tables("invert", N, False, (H, W), dev)
    gidx = (2 * torch.arange(B, device=dev) % N).to(torch.int32)          # 0, 2, 4, 6 -> 0, 90, 180, 270 degrees
    y = ops.canon_transform(x, gidx, th, fl, pad)
    for e in (0, 2, 4, 6):
        sel = gidx == e
        # rotate(x, -theta) with kornia's convention = clockwise by theta on the displayed image = rot90 with k = -theta / 90
        d = (y[sel] - torch.rot90(x[sel], k=-(e // 2), dims=(-2, -1))).abs()
        assert d.max().item() <= 1e-3 and d.pow(2).mean().sqrt().item() <= 1e-4, e
    mass = x.double().abs().sum((1, 2, 3))
    assert ((y.double().sum((1, 2, 3)) - x.double().sum((1, 2, 3))).abs() <= 1e-5 * mass).all()
    back = ops.invert_action(y, gidx, thi, fli, None)
    d = (back - x).abs()
    assert d.max().item() <= 2e-3 and d.pow(2).mean().sqrt().item() <= 2e-4


def test_resampling_is_linear_full_size(dev):
    """T(a x + b z) = a T(x) + b T(z) for every C8 element (bilinear resampling of an edge-padded frame is a linear map), on the
    headline batch: 2e-6 of the value scale, canonicalize and invert."""
    from equiadapt_amd import ops

    torch.manual_seed(101)
    B, C, H, W, N = 256, 3, 224, 224, 8
    x, z = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
    a, b = 0.75, -1.5
    pad = math.ceil(W * 0.5)
    th, fl = _tables("canonicalize", N, False, (H + 2 * pad, W + 2 * pad), dev)
    thi, fli, _ = _tables("invert", N, False, (H, W), dev)
    gidx = torch.randint(0, N, (B,), device=dev).to(torch.int32)
    lhs = ops.canon_transform(a * x + b * z, gidx, th, fl, pad)
    rhs = a * ops.canon_transform(x, gidx, th, fl, pad) + b * ops.canon_transform(z, gidx, th, fl, pad)
    assert (lhs - rhs).abs().max().item() <= 2e-6 * 8.0
    lhs = ops.invert_action(a * x + b * z, gidx, thi, fli, None)
    rhs = a * ops.invert_action(x, gidx, thi, fli, None) + b * ops.invert_action(z, gidx, thi, fli, None)
    assert (lhs - rhs).abs().max().item() <= 2e-6 * 8.0


def test_canonical_form_is_invariant_under_right_angle_rotations_full_size(dev):
    """The headline canonicalizer (ESCNN-shaped C8 network, 224 -> 96): rotating the INPUT by 90 degrees moves the chosen group
    element by two steps and leaves the canonical image unchanged (the network is exactly equivariant for right angles, so only
    summation order differs: images whose top-2 activation margin is below 1e-4 of the scale are exempt), and canonicalizing a
    canonical image chooses the identity."""
    can = bench.build_canonicalizer(dev)
    torch.manual_seed(102)
    B = 256
    # smooth images: white noise gives the random-init network orientation margins of ~1e-5 of the activations
    base = torch.randn(B, 3, 14, 14, device=dev)
    x = torch.nn.functional.interpolate(base, size=(224, 224), mode="bicubic", align_corners=False).contiguous()
    with torch.no_grad():
        y0 = can(x)
        acts0 = can.canonicalization_info_dict["group_activations"].clone()
        g0 = can.canonicalization_info_dict["group_index"].clone().long()
        xr = torch.rot90(x, k=1, dims=(-2, -1)).contiguous()
        y1 = can(xr)
        g1 = can.canonicalization_info_dict["group_index"].clone().long()
        acts1 = can.canonicalization_info_dict["group_activations"].clone()
    top2 = acts0.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4 * acts0.abs().max()
    assert int(clear.sum()) >= B // 2
    # equivariance of the activations: a rotation of the input by one right angle = a shift by 2 of the 8 orientations
    shift = ((g1 - g0) % 8)[clear]
    assert (shift == shift[0]).all() and int(shift[0]) in (2, 6)
    rolled = torch.roll(acts0, int(shift[0]), dims=1)
    assert (acts1 - rolled).abs().max().item() <= 1e-4 * acts0.abs().max().item()
    d = (y1 - y0)[clear].abs()
    assert d.max().item() <= 1e-3 and d.pow(2).mean().sqrt().item() <= 1e-4
    # idempotence for right-angle choices: the canonical image of such an input canonicalizes to itself with the identity
    right = clear & (g0 % 2 == 0)
    if int(right.sum()) > 0:
        with torch.no_grad():
            y2 = can(y0[right].contiguous())
            g2 = can.canonicalization_info_dict["group_index"].long()
        assert (g2 == 0).all()
        d = (y2 - y0[right]).abs()
        assert d.max().item() <= 1e-3 and d.pow(2).mean().sqrt().item() <= 1e-4


def test_mask_action_is_a_permutation_for_d4_full_size(dev):
    """96 uint8 masks of 1024 x 1024 (config 5), D4: every element moves pixels without interpolation -- the histogram of every
    mask is unchanged, the identity returns the input, and applying the inverse element returns the input."""
    from equiadapt_amd import ops
    from equiadapt_amd.images import geometry

    torch.manual_seed(103)
    P, S, N = 96, 1024, 4
    m = (torch.rand(P, S, S, device=dev) * 4).to(torch.uint8)
    # element e < 4: rotate by -90 e degrees; e >= 4: flip horizontally first, then the same rotation (how canonicalize_masks
    # builds its table when the group has reflections)
    ang = geometry.group_angles(N)
    rtheta = geometry.mask_rotation_table(torch.cat([-ang, -ang]).tolist(), (S, S)).to(dev)
    flags = torch.tensor([0] * N + [geometry.FLIP_SRC] * N, dtype=torch.int32, device=dev)
    eidx = (torch.arange(P, device=dev) % (2 * N)).to(torch.int32)
    out = ops.mask_action_nearest(m, eidx, rtheta, flags)
    for v in range(4):
        assert torch.equal((out == v).sum((1, 2)), (m == v).sum((1, 2))), v
    assert torch.equal(out[eidx == 0], m[eidx == 0])
    assert torch.equal(out[eidx == 4], torch.flip(m[eidx == 4], dims=(-1,)))
    # rotations: the inverse of element e is N - e; a flip followed by a rotation is an involution
    inv = torch.where(eidx < N, (N - eidx) % N, eidx).to(torch.int32)
    back = ops.mask_action_nearest(out, inv, rtheta, flags)
    assert torch.equal(back, m)


def test_pointcloud_canonical_form_is_rotation_invariant_full_size(dev):
    """2048 clouds of 1024 points (config 4): rotating the input by a random proper rotation leaves the canonical cloud unchanged
    (VNSmall is SO(3)-equivariant: R(Qx) = R(x) Q^T, so R(Qx) Qx = R(x) x) to 1e-3 of the coordinate scale for all but the few
    clouds whose kNN graph has a tie within rounding, and the rotation matrices stay orthonormal to 5e-4."""
    import equiadapt_amd as ea

    torch.manual_seed(104)
    hp = types.SimpleNamespace(n_knn=20, pooling="mean")
    can = ea.EquivariantPointcloudCanonicalization(ea.VNSmall(hp), hp).to(dev).eval()
    B, n = 2048, 1024
    x = torch.randn(B, 3, n, device=dev)
    q, r = torch.linalg.qr(torch.randn(B, 3, 3, device=dev))
    q = q * torch.sign(torch.diagonal(r, dim1=1, dim2=2)).unsqueeze(1)
    q[:, :, 0] *= torch.det(q).unsqueeze(1)                                  # proper rotations
    with torch.no_grad():
        y0 = can(x)
        R0 = can.canonicalization_info_dict["group_element"]["rotation"].clone()
        y1 = can(torch.bmm(q, x))
    err = (y1 - y0).abs().amax(dim=(1, 2))
    assert (err <= 1e-3 * 4.0).float().mean().item() >= 0.99, err.max().item()
    eye = torch.eye(3, device=dev).expand(B, 3, 3)
    assert (torch.bmm(R0, R0.transpose(1, 2)) - eye).abs().max().item() <= 5e-4
