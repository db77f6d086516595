"""CPU: property tests that guard the image part of the oracle, whose kornia/torchvision restatement cannot be pinned
to reference-generated vectors in this build (oracle/__init__.py, "PARITY UNPINNED").  Each property is independent of
how the restatement was written down: group structure, an fp64 pixel-space evaluation, torch.rot90, reference shapes."""
import math

import pytest
import torch

from oracle import image_ops as io
from oracle import nets as onets


def test_c4_rotation_is_rot90_counter_clockwise():
    torch.manual_seed(0)
    x = torch.randn(2, 3, 16, 16)
    for k in range(4):
        r = io.kornia_rotate(x, torch.tensor([90.0 * k] * 2))
        assert (r - torch.rot90(x, k, (-2, -1))).abs().max().item() < 5e-5  # +angle == counter-clockwise (fp32 trig: cos 90 = -4e-8)


def test_canonicalize_equals_fp64_clamp_gather_formulation():
    """pad(edge) -> flip blend -> rotate -> crop  ==  bilinear sample of a clamped, mirrored read (independent fp64)."""
    torch.manual_seed(1)
    x = torch.nn.functional.avg_pool2d(torch.randn(6, 2, 44, 36), 5, 1, 2)
    ang = torch.tensor([0.0, 45.0, 90.0, 135.0, 225.0, 315.0])
    ref = torch.tensor([0.0, 1.0, 1.0, 0.0, 1.0, 0.0])
    pad = math.ceil(36 * 0.5)
    got = io.canonicalize_images(x, ang, ref, (2, 44, 36))
    want = io.rotate_exact_fp64(x, -ang, pad=pad, crop_hw=(44, 36), pre_hflip=ref)
    assert (got.double() - want).abs().max().item() < 5e-5
    got = io.canonicalize_images(x[:, :1], ang, None, (1, 44, 36))  # grayscale: no pad / crop, zero corners
    want = io.rotate_exact_fp64(x[:, :1], -ang)
    assert (got.double() - want).abs().max().item() < 5e-5


def test_identity_element_is_identity():
    torch.manual_seed(2)
    x = torch.randn(3, 3, 20, 24)
    z = torch.zeros(3)
    # (the fp32 reference chain is not an exact identity at 0 deg: the normalise/inverse round trip costs ~1e-5)
    assert (io.canonicalize_images(x, z, None, (3, 20, 24)) - x).abs().max().item() < 5e-5
    assert (io.canonicalize_images(x, z, z, (3, 20, 24)) - x).abs().max().item() < 5e-5
    # invert: indicator 0 flips (reference convention), indicator 1 leaves alone
    f = torch.randn(3, 8, 12, 12)
    assert (io.invert_action(f, z, torch.ones(3), 4, 8, "scalar") - f).abs().max().item() < 5e-5
    assert (io.invert_action(f, z, z, 4, 8, "scalar") - f.flip(-1)).abs().max().item() < 5e-5


def test_invert_undoes_canonicalize_for_right_angles():
    torch.manual_seed(3)
    x = torch.randn(4, 3, 24, 24)
    ang = torch.tensor([0.0, 90.0, 180.0, 270.0])
    y = io.canonicalize_images(x, ang, None, (3, 24, 24))
    back = io.invert_action(y, ang, None, 4, 4, "scalar")
    assert (back - x).abs().max().item() < 1e-4


@pytest.mark.parametrize("N", [4, 8])
def test_regular_roll_composes_like_the_group(N):
    torch.manual_seed(4)
    f = torch.randn(1, 2 * N, 6, 6)
    ang = io.group_angles(N)
    for a in range(N):
        for b in range(N):
            one = io.invert_action(io.invert_action(f, ang[a : a + 1] * 0 + 0.0, None, N, N, "regular"), ang[b : b + 1] * 0, None, N, N, "regular")
            assert torch.allclose(one, f, atol=1e-5)
    # pure roll (rotation of a constant-per-channel map): element a then b == element (a + b) mod N
    g = torch.arange(2 * N, dtype=torch.float32).view(1, 2 * N, 1, 1).expand(1, 2 * N, 5, 5).contiguous()
    centre = lambda t: t[:, :, 2, 2]  # noqa: E731  (corners are zeroed by the rotation; the centre pixel is not)
    for a in range(N):
        for b in range(N):
            ab = io.invert_action(io.invert_action(g, ang[a : a + 1], None, N, N, "regular")[:, :, 2:3, 2:3].expand(1, 2 * N, 5, 5).contiguous(),
                                  ang[b : b + 1], None, N, N, "regular")
            direct = io.invert_action(g, ang[(a + b) % N].reshape(1), None, N, N, "regular")
            assert torch.allclose(centre(ab), centre(direct), atol=1e-4), (a, b)


def test_roll_by_gather_semantics():
    x = torch.arange(8, dtype=torch.float32).view(1, 1, 8, 1, 1)
    out = io.roll_by_gather(x, torch.tensor([3.0]))
    assert out.flatten().tolist() == [5, 6, 7, 0, 1, 2, 3, 4]  # out[g] = in[(g - 3) mod 8]
    assert io.roll_by_gather(x, torch.tensor([0.99999])).flatten().tolist() == list(range(8))  # .long() truncates


def test_orbit_contains_canonicalize_of_each_element():
    torch.manual_seed(5)
    x = torch.randn(2, 3, 16, 16)
    orbit = io.orbit_expand(x, 4, "rotation", 16)
    ang = io.group_angles(4)
    for e in range(4):
        one = io.canonicalize_images(x, ang[e].expand(2), None, (3, 16, 16))
        assert torch.equal(orbit[2 * e : 2 * e + 2], one)
    # reflected half: flip AFTER the rotation
    orb_d = io.orbit_expand(x, 4, "roto-reflection", 16)
    assert torch.equal(orb_d[8:], orb_d[:8].flip(-1))


def test_reference_shape_contracts():
    # tests/images/canonicalization/test_continuous_group.py:89-91
    x = torch.randn(1, 3, 64, 64)
    assert io.pre_canonicalization_transform(x, (3, 64, 64), 0.9, 32).shape == (1, 3, 32, 32)
    assert io.pre_canonicalization_transform(x[:, :1], (1, 64, 64), 0.9, 32).shape == (1, 1, 64, 64)  # grayscale: identity
    assert io.tv_resize_output_size((180, 240), 96) == (96, 128)
    assert io.tv_resize_output_size((180, 240), (32, 48)) == (32, 48)


@pytest.mark.parametrize("group_type", ["rotation", "roto-reflection"])
def test_network_oracle_is_equivariant_and_canonicalization_is_invariant(group_type):
    """Rotating the input by a group element cyclically shifts the activations; canonicalized images agree on the
    inscribed disc.  (C4: the bilinear filter rotation is exact, so this is tight.)"""
    import equiadapt_amd as ea

    torch.manual_seed(6)
    net = ea.CustomEquivariantNetwork((3, 24, 24), 4, 5, group_type, 4, 2, device="cpu")
    sd = net.state_dict()
    x = torch.randn(3, 3, 24, 24)
    a0 = onets.custom_equivariant_network(x, sd, group_type, 4, 2)
    a1 = onets.custom_equivariant_network(torch.rot90(x, 1, (-2, -1)), sd, group_type, 4, 2)
    assert torch.allclose(a1[:, :4], torch.roll(a0[:, :4], 1, dims=1), atol=1e-5)
    if group_type == "roto-reflection":
        assert torch.allclose(a1[:, 4:], torch.roll(a0[:, 4:], -1, dims=1), atol=1e-5)
    e0 = io.group_element_from_activations(a0, 4, group_type, 1.0, False)
    e1 = io.group_element_from_activations(a1, 4, group_type, 1.0, False)
    c0 = io.canonicalize_images(x, e0["rotation"], e0.get("reflection"), (3, 24, 24))
    c1 = io.canonicalize_images(torch.rot90(x, 1, (-2, -1)), e1["rotation"], e1.get("reflection"), (3, 24, 24))
    assert (c0 - c1).abs().max().item() < 1e-4


def test_mask_rotation_matches_rot90_for_right_angles():
    m = (torch.rand(2, 12, 12, generator=torch.Generator().manual_seed(7)) > 0.5).to(torch.uint8)
    assert torch.equal(io.rotate_masks(m, 90.0), torch.rot90(m, 1, (-2, -1)))
    assert torch.equal(io.rotate_masks(m, -90.0), torch.rot90(m, -1, (-2, -1)))
    b = torch.tensor([[2.0, 3.0, 6.0, 9.0]])
    assert torch.allclose(io.rotate_boxes(b.clone(), torch.tensor(0.0), 12), b)
    assert torch.allclose(io.flip_boxes(b.clone(), 12), torch.tensor([[6.0, 3.0, 10.0, 9.0]]))
