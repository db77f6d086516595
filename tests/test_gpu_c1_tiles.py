"""One-channel maps through group_action_c1_kernel (several tiles per block, csrc/group_action.hip) against the general kernel
(one tile per block: eqa_set_option(3, 0)) -- the output must be the same bits -- and against the CPU oracle.
Reference: discrete_group.py:204-238 (invert_canonicalization on a (B, 1, H, W) prediction), :387-481."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import image_ops as io  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    from equiadapt_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def _forms(fn):
    """fn() under eqa_set_option(3, v) for v = 0 (general kernel), 2, 4."""
    from equiadapt_amd import _lib

    lib = _lib.load()
    keep = lib.eqa_get_option(3)
    out = {}
    try:
        for v in (0, 2, 4):
            assert lib.eqa_set_option(3, v) == 0
            out[v] = fn()
    finally:
        lib.eqa_set_option(3, keep)
    return out


def _same_bits(forms):
    for v in (2, 4):
        assert torch.equal(forms[0], forms[v]), f"{v} tiles per block differ from the general kernel: max |d| = {(forms[0] - forms[v]).abs().max().item():.3e}"


def test_option_key_3_round_trips(dev):
    from equiadapt_amd import _lib

    lib = _lib.load()
    keep = lib.eqa_get_option(3)
    assert keep in (0, 2, 4)
    assert lib.eqa_set_option(3, 3) != 0 and lib.eqa_get_option(3) == keep
    assert lib.eqa_set_option(3, 2) == 0 and lib.eqa_get_option(3) == 2
    assert lib.eqa_set_option(3, keep) == 0


@pytest.mark.parametrize("B", [4, 32, 11])
def test_invert_of_config5_output_is_bit_identical(dev, B):
    """(B, 1, 1024, 1024), D4, the window hint of right-angle tables; B = 4 and 11: ragged image groups."""
    from equiadapt_amd import ops
    from equiadapt_amd.images.utils import device_tables

    th, fl, _ = device_tables("invert", 4, True, (1024, 1024), dev)
    torch.manual_seed(B)
    f = torch.randn(B, 1, 1024, 1024, device=dev)
    gidx = (torch.arange(B) % 8).to(dev, torch.int32)
    forms = _forms(lambda: ops.invert_action(f, gidx, th, fl, None))
    _same_bits(forms)
    # ... and it is the reference's map: a quarter turn / flip of a scalar map moves pixels, it does not blend them
    k = min(B, 8)
    rot, ref = _elements(4, True, gidx[:k].cpu().long())
    want = io.invert_action(f[:k].cpu(), rot, ref, 4, 8, "scalar")
    assert (forms[4][:k].cpu() - want).abs().max().item() <= 5e-4     # (white noise at a 1024-pixel frame, tests/test_gpu_parity.py:_pix_tol)


def _elements(N, refl, gidx):
    ang = io.group_angles(N)
    if not refl:
        return ang[gidx], None
    return torch.cat([ang, ang])[gidx], (gidx >= N).float()


@pytest.mark.parametrize("N,refl,shape,B", [(8, True, (250, 250), 70), (8, False, (224, 224), 96), (4, False, (300, 420), 40),
                                            (8, True, (131, 517), 64)])
def test_invert_odd_shapes_are_bit_identical(dev, N, refl, shape, B):
    """45-degree elements (47-wide windows, masked lanes), widths that are not a multiple of 4 (scalar stores), partial tiles, a
    ragged last image group, non-square frames (windows beyond the LDS budget -> the direct gather of the same arithmetic)."""
    from equiadapt_amd import ops
    from equiadapt_amd.images.utils import device_tables

    H, W = shape
    th, fl, _ = device_tables("invert", N, refl, (H, W), dev)
    G = 2 * N if refl else N
    torch.manual_seed(7)
    f = torch.randn(B, 1, H, W, device=dev)
    gidx = (torch.arange(B) % G).to(dev, torch.int32)
    _same_bits(_forms(lambda: ops.invert_action(f, gidx, th, fl, None)))


def test_gray_canonicalize_with_padding_matches_general_kernel_and_oracle(dev):
    """The canonicalizing transform of one-channel images (pad -> flip -> rotate -> crop) takes the same kernel."""
    from equiadapt_amd import ops
    from equiadapt_amd.images.utils import device_tables

    B, S, N = 96, 224, 8
    pad = math.ceil(S * 0.5)
    th, fl = device_tables("canonicalize", N, True, (S + 2 * pad, S + 2 * pad), dev)
    torch.manual_seed(3)
    x = torch.randn(B, 1, S, S)
    gidx = torch.arange(B) % 16
    xd, gd = x.to(dev), gidx.to(dev, torch.int32)
    forms = _forms(lambda: ops.canon_transform(xd, gd, th, fl, pad))
    _same_bits(forms)
    rot, ref = _elements(N, True, gidx)
    # oracle: the reference pads a one-channel image like any other when asked to; compare on the first 32 images
    want = io.canonicalize_images(x[:32].repeat(1, 3, 1, 1), rot[:32], ref[:32], (3, S, S))[:, :1]
    d = (forms[4][:32].cpu().double() - want.double()).abs()
    assert d.max().item() <= 2.8e-4 and d.pow(2).mean().sqrt().item() <= 2.4e-5


def test_orbit_mode_and_forced_direct_are_bit_identical(dev):
    """No group index (element-major orbit: n = e * B + b), and eqa_set_option(0, 1) (no LDS staging at all)."""
    from equiadapt_amd import _lib, ops
    from equiadapt_amd.images.utils import device_tables

    B, S = 12, 224
    th, fl = device_tables("orbit", 8, False, (S, S), dev)[:2]
    torch.manual_seed(5)
    x = torch.randn(B, 1, S, S, device=dev)
    _same_bits(_forms(lambda: ops.group_action(x, None, th, fl, None, 0, (S, S), (0, 0))))
    thi, fli, _ = device_tables("invert", 8, True, (S, S), dev)
    f = torch.randn(96, 1, S, S, device=dev)
    gidx = (torch.arange(96) % 16).to(dev, torch.int32)
    lib = _lib.load()
    staged = ops.invert_action(f, gidx, thi, fli, None)
    lib.eqa_set_option(0, 1)
    try:
        forms = _forms(lambda: ops.invert_action(f, gidx, thi, fli, None))
    finally:
        lib.eqa_set_option(0, 0)
    _same_bits(forms)
    assert torch.equal(forms[4], staged)
