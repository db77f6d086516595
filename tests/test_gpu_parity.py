"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Stated tolerances (SURVEY.md section 8d "Parity report"):
  * group index: bit-exact (int32) wherever the oracle's top-2 activation margin exceeds 1e-4;
  * resampled pixels: the fp32 reference itself sits up to ~4e-4 (max) / 4e-5 (rms) from exact arithmetic on
    unit-variance WHITE NOISE at a 448-px frame (normalised-grid rounding x pixel gradient), so
        max |hip - oracle| <= 1e-3   and   rms <= 1e-4        on white noise,
        max |hip - oracle| <= 5e-5                            on smooth images,
    and, against an fp64 evaluation of the same map, the HIP result must be no further away than the
    oracle is (x1.5 + 1e-6): the kernel is as exact as the reference path.
  * point-cloud rotation matrices 1e-4 abs, canonicalized coordinates 5e-4 abs (the reference's own R is
    orthonormal only to ~2e-4); the raw SO(3) action and Gram-Schmidt kernels: 1e-5.
"""
import math
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import image_ops as io  # noqa: E402
from oracle import pointcloud_ops as po  # noqa: E402

# Pixel tolerances on unit-variance white noise: 2 x the error measured at the headline's 448-pixel frame (1.4e-4 max / 1.2e-5 rms,
# profiles/r03/bench_n1.json self_check; the fp32 reference itself is 4e-4 / 4e-5 away from exact arithmetic there -- one ulp of a
# normalised coordinate is 3e-5 px, times a white-noise gradient).  The error grows with the frame (ulp of the coordinate x frame
# size): tests on larger frames scale the bound by frame / 448 (`_pix_tol`).  BASELINE.md's "1e-5 abs" is met on smooth images
# (SMOOTH_MAX); on white noise no fp32 implementation, the reference included, is that close to another one.
PIX_MAX, PIX_RMS, SMOOTH_MAX = 2.8e-4, 2.4e-5, 5e-5


# Distance to EXACT arithmetic (rot90, fp64 resampling) rather than to the oracle: the fp32 reference itself is 4e-4 away from it on
# white noise at a 448-pixel frame, and the product may be 1.5 x as far (the rule test_canonicalize_headline_shape_error_budget enforces)
EXACT_MAX = 6e-4


def _pix_tol(frame: int):
    k = max(1.0, frame / 448.0)
    return PIX_MAX * k, PIX_RMS * k


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    from equiadapt_amd import _lib

    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def _close(hip: torch.Tensor, ref: torch.Tensor, mx=PIX_MAX, rms=PIX_RMS):
    d = (hip.detach().cpu().double() - ref.double()).abs()
    assert d.max().item() <= mx, f"max err {d.max().item():.3e} > {mx}"
    assert d.pow(2).mean().sqrt().item() <= rms, f"rms err {d.pow(2).mean().sqrt().item():.3e} > {rms}"


def _elements(group_type, N, gidx):
    ang = io.group_angles(N)
    if group_type == "rotation":
        return ang[gidx], None
    return torch.cat([ang, ang])[gidx], (gidx >= N).float()


@pytest.mark.parametrize("group_type,N", [("rotation", 4), ("rotation", 8), ("roto-reflection", 4), ("roto-reflection", 8)])
@pytest.mark.parametrize("shape", [(3, 32, 32), (3, 64, 48), (2, 50, 70), (1, 28, 28), (5, 33, 31)])
def test_canonicalize_transform_matches_oracle(dev, group_type, N, shape):
    from equiadapt_amd import ops
    from equiadapt_amd.images.utils import device_tables
    import math

    C, H, W = shape
    G = N if group_type == "rotation" else 2 * N
    torch.manual_seed(1)
    B = 2 * G + 1
    x = torch.randn(B, C, H, W)
    gidx = torch.arange(B) % G
    rot, ref = _elements(group_type, N, gidx)
    want = io.canonicalize_images(x, rot, ref, shape)
    pad = 0 if C == 1 else math.ceil(W * 0.5)
    theta, flags = device_tables("canonicalize", N, group_type != "rotation", (H + 2 * pad, W + 2 * pad), dev)
    got = ops.canon_transform(x.to(dev), gidx.to(dev, torch.int32), theta, flags, pad)
    _close(got, want)


def test_canonicalize_headline_shape_error_budget(dev):
    """224x224x3 C8 (BASELINE config 2): vs oracle, and vs fp64 exact arithmetic."""
    from equiadapt_amd import ops
    from equiadapt_amd.images.utils import device_tables

    torch.manual_seed(0)
    B = 16
    x = torch.randn(B, 3, 224, 224)
    gidx = torch.arange(B) % 8
    rot = io.group_angles(8)[gidx]
    theta, flags = device_tables("canonicalize", 8, False, (448, 448), dev)
    got = ops.canon_transform(x.to(dev), gidx.to(dev, torch.int32), theta, flags, 112).cpu()
    want = io.canonicalize_images(x, rot, None, (3, 224, 224))
    _close(got, want)
    exact = io.rotate_exact_fp64(x, -rot, pad=112, crop_hw=(224, 224))
    e_hip = (got.double() - exact).abs().max().item()
    e_ref = (want.double() - exact).abs().max().item()
    assert e_hip <= 1.5 * e_ref + 1e-6, (e_hip, e_ref)
    # smooth images: tight
    xs = torch.nn.functional.avg_pool2d(x, 9, 1, 4)
    got = ops.canon_transform(xs.to(dev), gidx.to(dev, torch.int32), theta, flags, 112)
    _close(got, io.canonicalize_images(xs, rot, None, (3, 224, 224)), mx=SMOOTH_MAX, rms=1e-5)


@pytest.mark.parametrize("group_type,N", [("rotation", 4), ("rotation", 8), ("roto-reflection", 4)])
@pytest.mark.parametrize("rep", ["scalar", "regular"])
@pytest.mark.parametrize("hw", [(32, 32), (40, 56), (224, 224)])
def test_invert_action_matches_oracle(dev, group_type, N, rep, hw):
    from equiadapt_amd.images.utils import get_action_on_image_features

    G = N if group_type == "rotation" else 2 * N
    H, W = hw
    torch.manual_seed(3)
    B = G + 3 if H < 100 else 4
    C = 2 * G if rep == "regular" else 3
    f = torch.randn(B, C, H, W)
    gidx = (torch.arange(B) * 3 + 1) % G
    rot, ref = _elements(group_type, N, gidx)
    want = io.invert_action(f, rot, ref, N, G, rep)
    element = {"rotation": rot.to(dev)}
    if ref is not None:
        element["reflection"] = ref.to(dev)
    got = get_action_on_image_features(f.to(dev), {"num_rotations": N, "num_group": G}, element, rep)
    _close(got, want)
    # and with the explicit int index (what the canonicalizer passes)
    element["group_index"] = gidx.to(dev, torch.int32)
    got2 = get_action_on_image_features(f.to(dev), {"num_rotations": N, "num_group": G}, element, rep)
    assert torch.equal(got, got2)


def test_invert_errors_like_reference(dev):
    from equiadapt_amd.images.utils import get_action_on_image_features

    f = torch.zeros(1, 8, 8, 8, device=dev)
    el = {"rotation": torch.zeros(1, device=dev)}
    info = {"num_rotations": 4, "num_group": 4}
    with pytest.raises(NotImplementedError):
        get_action_on_image_features(f, info, el, "vector")
    with pytest.raises(ValueError):
        get_action_on_image_features(f, info, el, "tensor")
    with pytest.raises(AssertionError):
        get_action_on_image_features(torch.zeros(1, 6, 8, 8, device=dev), info, el, "regular")


@pytest.mark.parametrize("group_type,N", [("rotation", 4), ("roto-reflection", 4), ("rotation", 8)])
def test_orbit_expand_matches_oracle(dev, group_type, N):
    from equiadapt_amd import ops
    from equiadapt_amd.images.utils import device_tables

    torch.manual_seed(4)
    x = torch.randn(3, 3, 32, 32)
    want = io.orbit_expand(x, N, group_type, 32)
    theta, flags = device_tables("orbit", N, group_type != "rotation", (64, 64), dev)
    got = ops.orbit_expand(x.to(dev), theta, flags, 16)
    _close(got, want)


def test_golden_image_fixtures(dev, golden):
    """Committed restatement-generated fixtures (labelled parity-unpinned) reproduce on the GPU."""
    import equiadapt_amd as ea
    from equiadapt_amd import ops
    from equiadapt_amd.images.utils import device_tables, get_action_on_image_features

    g = golden("images_restatement.pt")
    x = g["x"].to(dev)
    for key, N, refl in (("c8", 8, False), ("d4", 4, True)):
        c = g[key]
        gidx = c["gidx"].to(dev, torch.int32)
        theta, flags = device_tables("canonicalize", N, refl, (64, 64), dev)
        _close(ops.canon_transform(x, gidx, theta, flags, 16), c["canon"])
        el = {"group_index": gidx, "rotation": None}
        if refl:
            el["reflection"] = None
        info = {"num_rotations": N, "num_group": 8}
        _close(get_action_on_image_features(c["f"].to(dev), info, el, "regular"), c["invert_regular"])
        _close(get_action_on_image_features(c["f"][:, :3].contiguous().to(dev), info, el, "scalar"), c["invert_scalar"])
    theta, flags = device_tables("orbit", 4, True, (64, 64), dev)
    _close(ops.orbit_expand(x[:2].contiguous(), theta, flags, 16), g["orbit_d4"])
    theta, flags = device_tables("canonicalize", 4, False, (32, 32), dev)
    _close(ops.canon_transform(x[:, :1].contiguous(), torch.arange(4, device=dev, dtype=torch.int32), theta, flags, 0), g["gray_c4"])
    _close(ea.rotate_masks(g["masks"]["in"].to(dev), -45.0).float(), g["masks"]["rot_m45"].float(), mx=0, rms=0)
    _close(ea.rotate_masks(g["masks"]["in"].to(dev), 90.0).float(), g["masks"]["rot_90"].float(), mx=0, rms=0)


def test_lds_and_direct_paths_agree(dev):
    from equiadapt_amd import _lib, ops
    from equiadapt_amd.images.utils import device_tables

    torch.manual_seed(5)
    x = torch.randn(9, 3, 96, 96, device=dev)
    gidx = (torch.arange(9, device=dev) % 8).to(torch.int32)
    theta, flags = device_tables("canonicalize", 8, False, (192, 192), dev)
    a = ops.canon_transform(x, gidx, theta, flags, 48)
    lib = _lib.load()
    lib.eqa_set_option(0, 1)
    try:
        b = ops.canon_transform(x, gidx, theta, flags, 48)
    finally:
        lib.eqa_set_option(0, 0)
    # same arithmetic, but the compiler contracts the four multiply-adds differently in the two branches
    assert (a - b).abs().max().item() <= 2e-6


def test_c4_is_rot90_and_identity_element(dev):
    """Size-independent properties at the full BASELINE size (B=64, 224x224x3)."""
    from equiadapt_amd import ops
    from equiadapt_amd.images.utils import device_tables

    torch.manual_seed(6)
    x = torch.randn(64, 3, 224, 224, device=dev)
    theta, flags = device_tables("canonicalize", 4, False, (448, 448), dev)
    for k in range(4):
        gidx = torch.full((64,), k, device=dev, dtype=torch.int32)
        y = ops.canon_transform(x, gidx, theta, flags, 112)
        # canonicalize rotates by -angle: element k undoes a +k*90 deg rotation == rot90(k=-k)
        # white noise: the reference's own cos(90 deg) = -4.4e-8 and grid rounding move samples by ~3e-5 px
        assert (y - torch.rot90(x, -k, (-2, -1))).abs().max().item() <= EXACT_MAX
    # invert(canonicalize(x)) == x on the inscribed disc (C8, scalar features)
    th_c, fl_c = device_tables("canonicalize", 8, False, (448, 448), dev)
    th_i, fl_i, _ = device_tables("invert", 8, False, (224, 224), dev)
    gidx = (torch.arange(64, device=dev) % 8).to(torch.int32)
    yy, xx = torch.meshgrid(torch.arange(224.0, device=dev), torch.arange(224.0, device=dev), indexing="ij")
    # a smooth analytic image (curvature ~2.5e-3 / px^2): bilinear resampling error ~ curvature / 8 per pass
    base = torch.sin(0.05 * xx + 0.3) * torch.cos(0.04 * yy) + 0.01 * xx
    xs = (base[None, None] * torch.linspace(0.5, 1.5, 64 * 3, device=dev).view(64, 3, 1, 1)).contiguous()
    back = ops.invert_action(ops.canon_transform(xs, gidx, th_c, fl_c, 112), gidx, th_i, fl_i, None)
    disc = ((yy - 111.5) ** 2 + (xx - 111.5) ** 2) < 105.0**2
    err = ((back - xs).abs() * disc).amax(dim=(1, 2, 3))
    assert err[gidx.long() % 2 == 0].max().item() <= 1e-4   # multiples of 90 deg: a permutation
    assert err.max().item() <= 2e-3                          # 45 deg: two bilinear passes of a smooth image


def test_group_pool_argmax(dev):
    from equiadapt_amd import ops

    torch.manual_seed(7)
    for (B, Cf, G, Hf, Wf) in [(5, 8, 8, 28, 28), (3, 6, 4, 13, 11), (2, 32, 8, 84, 84), (70, 3, 16, 9, 9)]:
        fm = torch.randn(B, Cf, G, Hf, Wf)
        fm += torch.randn(B, 1, G, 1, 1) * 0.05  # give the orientations distinct means
        act, gidx = ops.group_pool_argmax(fm.to(dev))
        want = io.group_pool(fm)
        assert torch.allclose(act.cpu(), want, atol=1e-6, rtol=1e-5)
        top2 = want.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4
        assert clear.any()
        assert torch.equal(gidx.cpu().long()[clear], want.argmax(-1)[clear])
        # exactness against an fp64 reduction
        exact = fm.double().mean(dim=(1, 3, 4))
        assert (act.cpu().double() - exact).abs().max().item() <= 1e-7


def test_group_argmax_ties_and_nan(dev):
    from equiadapt_amd import ops

    a = torch.tensor([[0.0, 2.0, 2.0, 1.0], [5.0, 5.0, 5.0, 5.0], [-1.0, -3.0, -1.0, -2.0], [0.0, float("nan"), 9.0, float("nan")]])
    got = ops.group_argmax(a.to(dev)).cpu().long()
    assert got.tolist() == torch.argmax(a, dim=-1).tolist() == [1, 0, 0, 1]
    torch.manual_seed(8)
    r = torch.randn(1000, 16)
    assert torch.equal(ops.group_argmax(r.to(dev)).cpu().long(), r.argmax(-1))


def test_so3_rotate_and_gram_schmidt(dev, golden):
    from equiadapt_amd import ops

    g = golden("gram_schmidt.pt")
    out = ops.gram_schmidt(g["batch_in"].to(dev)).cpu()
    assert torch.allclose(out, g["batch_out"], atol=1e-5, rtol=0)
    kat = ops.gram_schmidt(g["kat_in"].to(dev)).cpu()
    assert torch.allclose(kat[0][0][0], torch.tensor(0.5740), atol=1e-4)  # the reference's own KAT
    torch.manual_seed(9)
    for N in (1024, 1000, 7):
        x = torch.randn(6, 3, N)
        R = po.gram_schmidt(torch.randn(6, 3, 3))
        assert torch.allclose(ops.so3_rotate(x.to(dev), R.to(dev)).cpu(), po.canonicalize_pointcloud(x, R), atol=1e-5, rtol=0)
        assert torch.allclose(ops.so3_rotate(x.to(dev), R.to(dev), transpose=True).cpu(),
                              torch.bmm(R.transpose(1, 2), x), atol=1e-5, rtol=0)


def test_pointcloud_canonicalizer_matches_reference_golden(dev, golden):
    import equiadapt_amd as ea

    g = golden("pointcloud.pt")
    for pooling in ("mean", "max"):
        c = g[pooling]
        hp = types.SimpleNamespace(n_knn=20, pooling=pooling)
        net = ea.VNSmall(hp)
        net.load_state_dict(c["state"])
        can = ea.EquivariantPointcloudCanonicalization(net, hp).to(dev).eval()
        with torch.no_grad():
            xc = can(c["x"].to(dev))
        R = can.canonicalization_info_dict["group_element_matrix_representation"].cpu()
        # network output agrees to ~1e-6 (kNN sets identical); Gram-Schmidt on 0.1-norm vectors amplifies ~25x
        assert torch.allclose(R, c["rotation"], atol=1e-4, rtol=0), pooling
        assert torch.allclose(xc.cpu(), c["x_canonicalized"], atol=5e-4, rtol=0), pooling
        assert torch.allclose(can.get_prior_regularization_loss().cpu(), c["prior_loss"], atol=1e-5)
        assert torch.allclose(can.get_identity_metric().cpu(), c["identity_metric"], atol=1e-5)
    with pytest.raises(NotImplementedError):
        can.invert_canonicalization(xc)


@pytest.mark.parametrize("group_type", ["rotation", "roto-reflection"])
def test_group_equivariant_canonicalizer_end_to_end(dev, group_type):
    """Module level: same network weights on both sides; index exact, images within tolerance."""
    import equiadapt_amd as ea
    from oracle import nets as onets

    torch.manual_seed(10)
    N = 4
    net = ea.CustomEquivariantNetwork((3, 32, 32), 8, 5, group_type, N, 2, device="cpu")
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=0.8, resize_shape=32)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 64, 64)).to(dev).eval()
    x = torch.randn(12, 3, 64, 64)
    with torch.no_grad():
        y = can(x.to(dev))
        acts = can.canonicalization_info_dict["group_activations"].cpu()
        gidx = can.canonicalization_info_dict["group_index"].cpu().long()
        f = torch.randn(12, 2 * can.num_group, 64, 64)
        inv = can.invert_canonicalization(f.to(dev))
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    xin = io.pre_canonicalization_transform(x, (3, 64, 64), 0.8, 32)
    acts_ref = onets.custom_equivariant_network(xin, sd, group_type, N, 2)
    assert torch.allclose(acts, acts_ref, atol=2e-5, rtol=1e-4)
    top2 = acts_ref.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(gidx[clear], acts_ref.argmax(-1)[clear])
    el = io.group_element_from_activations(acts, N, group_type, 1.0, training=False)
    assert torch.equal(can.canonicalization_info_dict["group_element"]["rotation"].cpu(), el["rotation"])
    _close(y, io.canonicalize_images(x, el["rotation"], el.get("reflection"), (3, 64, 64)))
    _close(inv, io.invert_action(f, el["rotation"], el.get("reflection"), N, can.num_group, "regular"))
    assert torch.allclose(can.get_prior_regularization_loss().cpu(), io.prior_regularization_loss(acts), atol=1e-6)
    assert torch.equal(can.get_identity_metric().cpu(), io.identity_metric(acts))


def test_window_sums_matches_unfold(dev):
    from equiadapt_amd import ops

    torch.manual_seed(11)
    for (B, C, H, W, k) in [(3, 5, 20, 24, 5), (2, 8, 88, 88, 5), (4, 3, 9, 9, 1), (1, 2, 17, 13, 3), (2, 4, 30, 26, 9), (1, 3, 19, 40, 10), (2, 6, 15, 15, 8)]:
        x = torch.randn(B, C, H, W)
        scale, shift = torch.rand(C) + 0.5, torch.randn(C) * 0.3
        got = ops.window_sums(x.to(dev), k, scale.to(dev), shift.to(dev), relu=True).cpu()
        a = torch.relu(x.double() * scale.double()[None, :, None, None] + shift.double()[None, :, None, None])
        OH, OW = H - k + 1, W - k + 1
        want = torch.stack([torch.stack([a[:, :, u:u + OH, v:v + OW].sum(dim=(2, 3)) for v in range(k)], -1) for u in range(k)], -2)
        assert torch.allclose(got, want, atol=1e-6 * H * W, rtol=1e-9)
        got = ops.window_sums(x.to(dev), k).cpu()  # no affine, no relu
        want = torch.stack([torch.stack([x.double()[:, :, u:u + OH, v:v + OW].sum(dim=(2, 3)) for v in range(k)], -1) for u in range(k)], -2)
        assert torch.allclose(got, want, atol=1e-6 * H * W, rtol=1e-9)


def test_window_sums_channels_last_few_channels(dev):
    """eqa_window_sums_nhwc with fewer than 64 channel quads: a wave instruction covers several pixels (C = 8, 16, 32, 64, 128), the
    general path (C = 48: 12 quads do not divide 64; C = 512: two trips), k = 1 (the 1 x 1 tail of the CIFAR-shaped network),
    3 and 5, widths that do not divide the pixel step -- against fp64 unfold sums."""
    from equiadapt_amd import ops

    torch.manual_seed(13)
    for (B, C, H, W, k) in [(3, 32, 28, 28, 1), (2, 8, 30, 37, 3), (2, 16, 12, 9, 5), (1, 64, 20, 21, 5), (2, 128, 11, 50, 3), (2, 48, 14, 14, 3),
                            (1, 512, 10, 12, 1), (5, 32, 9, 200, 5), (3, 64, 48, 48, 9), (2, 16, 17, 23, 9), (2, 32, 40, 19, 10),
                            (2, 24, 30, 30, 7), (1, 8, 15, 16, 8)]:
        x = torch.randn(B, C, H, W).to(dev).contiguous(memory_format=torch.channels_last)
        scale, shift = (torch.rand(C) + 0.5).to(dev), (torch.randn(C) * 0.3).to(dev)
        assert not x.is_contiguous() or C == 1
        got = ops.window_sums(x, k, scale, shift, relu=True).cpu()
        a = torch.relu(x.double().cpu() * scale.double().cpu()[None, :, None, None] + shift.double().cpu()[None, :, None, None])
        OH, OW = H - k + 1, W - k + 1
        want = torch.stack([torch.stack([a[:, :, u:u + OH, v:v + OW].sum(dim=(2, 3)) for v in range(k)], -1) for u in range(k)], -2)
        assert torch.allclose(got, want, atol=2e-6 * H * W, rtol=1e-9), (B, C, H, W, k)


@pytest.mark.parametrize("group_type", ["rotation", "roto-reflection"])
def test_linear_tail_fast_path_equals_conv_path(dev, group_type):
    """Inference fast path (window sums instead of the last conv, BN folded) vs the plain module path and the oracle."""
    import equiadapt_amd as ea
    from oracle import nets as onets

    torch.manual_seed(12)
    x = torch.randn(5, 3, 40, 40)
    for net in (ea.ESCNNEquivariantNetwork((3, 40, 40), 4, 5, group_type, 4, 3),
                ea.CustomEquivariantNetwork((3, 40, 40), 4, 5, group_type, 4, 2, device="cpu"),
                ea.CustomEquivariantNetwork((3, 40, 40), 4, 5, group_type, 4, 1, device="cpu")):
        for m in net.modules():  # non-trivial eval-mode statistics
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.normal_(0.1, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.2)
        net = net.to(dev).eval()
        with torch.no_grad():
            fast = net(x.to(dev)).cpu()
        with torch.enable_grad():
            slow = net(x.to(dev)).detach().cpu()  # grad mode -> conv + group_pool path
        assert torch.allclose(fast, slow, atol=2e-6, rtol=1e-4), type(net).__name__
        sd = {k: v.cpu() for k, v in net.state_dict().items()}
        if isinstance(net, ea.ESCNNEquivariantNetwork):
            want = onets.escnn_like_network(x, sd, group_type, 4, 3, 4)
        else:
            want = onets.custom_equivariant_network(x, sd, group_type, 4, len(net.eqv_network) // 2 + 1)
        assert torch.allclose(fast, want, atol=2e-5, rtol=1e-3), type(net).__name__


def test_fused_vnsmall_matches_reference_golden_and_op_path(dev, golden):
    """eqa_vnsmall_fwd (one kernel) vs the reference-generated vectors and vs the op-by-op module path."""
    import equiadapt_amd as ea

    c = golden("pointcloud.pt")["mean"]
    hp = types.SimpleNamespace(n_knn=20, pooling="mean")
    net = ea.VNSmall(hp)
    net.load_state_dict(c["state"])
    net = net.to(dev).eval()
    with torch.no_grad():
        fused = net(c["x"].to(dev)).cpu()
    assert torch.allclose(fused, c["vnsmall_out"], atol=2e-6, rtol=1e-4)
    torch.manual_seed(13)
    for (B, N) in [(3, 1024), (2, 300), (5, 64), (1, 20)]:
        x = torch.randn(B, 3, N, device=dev)
        with torch.no_grad():
            fused = net(x)
        with torch.enable_grad():
            slow = net(x).detach()  # grad mode -> op-by-op torch path
        assert torch.allclose(fused, slow, atol=3e-6, rtol=1e-4), (B, N)
    # equivariance: rotating the cloud rotates the three output vectors
    x = torch.randn(4, 3, 512, device=dev)
    R = torch.linalg.qr(torch.randn(4, 3, 3, device=dev)).Q
    with torch.no_grad():
        v, vr = net(x), net(torch.bmm(R, x))
    assert torch.allclose(vr, torch.bmm(v, R.transpose(1, 2)), atol=1e-5)


def test_fused_vnsmall_max_pooling_matches_reference_golden(dev, golden):
    """pooling="max" through the fused kernel (eqa_vnsmall_fwd, pooling 1): VNMaxPool's argmax over the 20 edges of <x, W_p x>
    (vector_neuron_layers.py:349-364) against the reference-generated vectors, the whole canonicalizer (Gram-Schmidt, rotate,
    losses) against the reference's, and against the op-by-op module path.  An argmax over near-ties can move on a last-bit
    difference of a score; one moved pick changes the mean over N points by |dx| / N, hence the 2e-5 bound."""
    import equiadapt_amd as ea

    c = golden("pointcloud.pt")["max"]
    hp = types.SimpleNamespace(n_knn=20, pooling="max")
    net = ea.VNSmall(hp)
    net.load_state_dict(c["state"])
    net = net.to(dev).eval()
    assert net.packed_parameters().numel() == 1751
    with torch.no_grad():
        fused = net(c["x"].to(dev)).cpu()
    err = (fused - c["vnsmall_out"]).abs().max().item()
    assert err <= 2e-5 * max(c["vnsmall_out"].abs().max().item(), 1.0), err
    can = ea.EquivariantPointcloudCanonicalization(net, hp).to(dev).eval()
    with torch.no_grad():
        xc = can(c["x"].to(dev)).cpu()
    R = can.canonicalization_info_dict["group_element_matrix_representation"].cpu()
    assert torch.allclose(R, c["rotation"], atol=1e-4) and torch.allclose(xc, c["x_canonicalized"], atol=5e-4)
    assert torch.allclose(can.get_prior_regularization_loss().cpu(), c["prior_loss"], atol=1e-4)
    torch.manual_seed(14)
    for (B, N) in [(3, 1024), (2, 300), (1, 20)]:
        x = torch.randn(B, 3, N, device=dev)
        with torch.no_grad():
            fused = net(x)
        with torch.enable_grad():
            slow = net(x).detach()  # grad mode -> op-by-op torch path
        assert (fused - slow).abs().max().item() <= 2e-5 * max(slow.abs().max().item(), 1.0), (B, N, (fused - slow).abs().max().item())
    x = torch.randn(4, 3, 512, device=dev)
    Rr = torch.linalg.qr(torch.randn(4, 3, 3, device=dev)).Q
    with torch.no_grad():
        v, vr = net(x), net(torch.bmm(Rr, x))
    assert torch.allclose(vr, torch.bmm(v, Rr.transpose(1, 2)), atol=2e-5)


def test_crop_resize_aa_matches_torch_interpolate(dev):
    """I1: eqa_crop_resize_aa vs torchvision semantics (CenterCrop -> F.interpolate(antialias=True)) on the CPU."""
    from equiadapt_amd import ops
    from equiadapt_amd.images import geometry

    torch.manual_seed(14)
    # the last three use the wide-filter (LDS row staging) kernel: 8x, 6x (even stride, padded rows) and 5x (odd stride)
    for (H, W, ratio, size) in [(224, 224, 0.8, 96), (64, 64, 0.9, 32), (50, 70, 0.8, (24, 40)), (33, 33, 1.0, 17),
                                (1024, 1024, 1.0, 128), (400, 300, 0.9, (60, 45)), (320, 320, 1.0, 64),
                                # narrow filters over 16-byte aligned rows: the LDS-staged band kernel (incl. up-sampling, output
                                # rows wider than a block, a window that starts off the 16-byte grid)
                                (96, 100, 0.75, 40), (256, 256, 1.0, (300, 300)), (64, 512, 1.0, (32, 300)), (180, 184, 0.95, (90, 77))]:
        x = torch.randn(3, 3, H, W)
        want = io.pre_canonicalization_transform(x, (3, H, W), ratio, size)
        import math
        crop = (math.ceil(H * ratio), math.ceil(W * ratio))
        out_hw = io.tv_resize_output_size(crop, size)
        tabs = tuple(v.to(dev) if isinstance(v, torch.Tensor) else v for v in geometry.aa_resize_tables((H, W), crop, out_hw))
        got = ops.crop_resize_aa(x.to(dev), tabs, out_hw).cpu()
        assert got.shape == want.shape
        assert (got - want).abs().max().item() <= 2e-6, (H, W, ratio, size)


def test_mask_action_nearest_bit_exact(dev, golden):
    """I6: uint8 nearest rotation -- bit-exact against the oracle's torchvision restatement."""
    import equiadapt_amd as ea
    from equiadapt_amd.images.utils import canonicalize_masks

    g = torch.Generator().manual_seed(15)
    for (H, W) in [(32, 32), (40, 56), (224, 224)]:
        m = (torch.rand(5, H, W, generator=g) > 0.5).to(torch.uint8) * 255
        for ang in (-45.0, 90.0, 135.0, -270.0, 0.0, 315.0):
            want = io.rotate_masks(m, ang)
            got = ea.rotate_masks(m.to(dev), ang).cpu()
            assert torch.equal(got, want), (H, W, ang)
    # batched, per-sample element, with the reference's "flip every target" behaviour for D_n
    masks = [(torch.rand(n, 48, 48, generator=g) > 0.5).to(torch.uint8) for n in (2, 0, 3, 1)]
    gidx = torch.tensor([1, 6, 3, 4])
    got = canonicalize_masks([m.to(dev) for m in masks], gidx.to(dev, torch.int32), 4, flip_all=True)
    ang = io.group_angles(4)
    for t, m in enumerate(masks):
        want = io.rotate_masks(io.flip_masks(m), -ang[gidx[t] % 4].item()) if m.shape[0] else m
        assert torch.equal(got[t].cpu(), want), t


def test_mask_action_u8_dword_staged_kernel_and_plane_table_bit_exact(dev):
    """The uint8 mask kernel with dword staging / 16-byte stores (widths that are multiples of 16) and its plane-pointer-table
    entry point (eqa_mask_action_nearest_planes: per-sample mask tensors, no concatenation): bit-exact against the oracle's
    torchvision restatement for every element of C8 and D4, square and non-square planes, ragged tiles, and against the
    byte-staged kernel it replaces."""
    import equiadapt_amd as ea
    from equiadapt_amd import _lib, ops
    from equiadapt_amd.images.utils import canonicalize_masks

    g = torch.Generator().manual_seed(16)
    for (H, W) in [(64, 64), (80, 112), (144, 48), (256, 256)]:
        m = (torch.rand(3, H, W, generator=g) * 255).to(torch.uint8)
        for ang in (-45.0, 90.0, 135.0, -270.0, 0.0, 315.0, 180.0):
            want = io.rotate_masks(m, ang)
            got = ea.rotate_masks(m.to(dev), ang)
            assert torch.equal(got.cpu(), want), (H, W, ang)
            _lib.load().eqa_set_option(0, 1)              # the row-per-block kernel without staging
            try:
                old = ea.rotate_masks(m.to(dev), ang)
            finally:
                _lib.load().eqa_set_option(0, 0)
            assert torch.equal(old, got), (H, W, ang)
    for (N, flip, S) in [(8, False, 128), (4, True, 96)]:
        masks = [(torch.rand(n, S, S, generator=g) > 0.5).to(torch.uint8) for n in (2, 0, 3, 1, 4)]
        G = 2 * N if flip else N
        gidx = torch.randint(0, G, (5,), generator=g)
        calls = []
        orig = ops.mask_action_nearest_planes
        ops.mask_action_nearest_planes = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            got = canonicalize_masks([m.to(dev) for m in masks], gidx.to(dev, torch.int32), N, flip_all=flip)
        finally:
            ops.mask_action_nearest_planes = orig
        assert calls, "the plane-pointer path was not taken"
        ang = io.group_angles(N)
        for t, m in enumerate(masks):
            src = io.flip_masks(m) if flip else m
            want = io.rotate_masks(src, -ang[gidx[t] % N].item()) if m.shape[0] else m
            assert torch.equal(got[t].cpu(), want), (N, flip, t)


def test_canonicalize_with_targets(dev):
    """Targets branch (reference discrete_group.py:217-238): boxes and masks follow the image."""
    import equiadapt_amd as ea

    torch.manual_seed(16)
    net = ea.CustomEquivariantNetwork((3, 32, 32), 4, 5, "rotation", 4, 1, device="cpu")
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=0.8, resize_shape=32)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 64, 64)).to(dev).eval()
    x = torch.randn(3, 3, 64, 64)
    targets = [{"boxes": torch.tensor([[4.0, 8.0, 20.0, 30.0]]).to(dev), "masks": (torch.rand(2, 64, 64) > 0.5).to(torch.uint8).to(dev)}
               for _ in range(3)]
    ref_t = [{k: v.clone().cpu() for k, v in t.items()} for t in targets]
    with torch.no_grad():
        y, out_t = can(x.to(dev), targets)
    rot = can.canonicalization_info_dict["group_element"]["rotation"].cpu()
    for t in range(3):
        assert torch.equal(out_t[t]["masks"].cpu(), io.rotate_masks(ref_t[t]["masks"], -rot[t].item()))
        assert torch.allclose(out_t[t]["boxes"].cpu(), io.rotate_boxes(ref_t[t]["boxes"], rot[t], 64), atol=1e-4)
    _close(y, io.canonicalize_images(x, rot, None, (3, 64, 64)))
    # reflections, a different number of boxes per image (one image has none): all boxes go through one batched pass
    net = ea.CustomEquivariantNetwork((3, 32, 32), 4, 5, "roto-reflection", 4, 1, device="cpu")
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 64, 64)).to(dev).eval()
    nbox = [3, 0, 1, 5]
    x = torch.randn(4, 3, 64, 64)
    boxes = [torch.rand(n, 4) * 30 + torch.tensor([0.0, 0.0, 30.0, 30.0]) for n in nbox]
    targets = [{"boxes": b.clone().to(dev), "masks": (torch.rand(1, 64, 64) > 0.5).to(torch.uint8).to(dev)} for b in boxes]
    held = [t["boxes"] for t in targets]  # the caller's tensors: the reference flips them in place
    with torch.no_grad():
        y, out_t = can(x.to(dev), targets)
    rot = can.canonicalization_info_dict["group_element"]["rotation"].cpu()
    for t in range(4):
        flipped = io.flip_boxes(boxes[t].clone(), 64)
        assert out_t[t]["boxes"].shape == (nbox[t], 4)
        assert torch.allclose(out_t[t]["boxes"].cpu(), io.rotate_boxes(flipped, rot[t], 64).reshape(nbox[t], 4), atol=1e-4)
        assert torch.allclose(held[t].cpu(), flipped)


@pytest.mark.parametrize("group_type,N", [("roto-reflection", 4), ("rotation", 8)])
def test_optimized_canonicalizer_end_to_end(dev, group_type, N):
    """OptimizedGroupEquivariantImageCanonicalization (I8 + I10): orbit kernel -> ConvNetwork -> cosine similarity ->
    element -> canonicalize, against the oracle's op sequence with the same weights (eval mode)."""
    import copy

    import equiadapt_amd as ea

    torch.manual_seed(17)
    G = N if group_type == "rotation" else 2 * N
    net = ea.ConvNetwork((3, 32, 32), out_channels=8, kernel_size=3, num_layers=2, out_vector_size=16)
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=0.8, resize_shape=32, group_type=group_type, num_rotations=N,
                               artifact_err_wt=0.0, learn_ref_vec=False)
    can = ea.OptimizedGroupEquivariantImageCanonicalization(net, hp, (3, 64, 64))
    cpu_net, ref_vec = copy.deepcopy(net).eval(), can.reference_vector.detach().clone()
    sd_net = {k: v.clone() for k, v in net.state_dict().items()}
    can = can.to(dev).eval()
    x = torch.randn(6, 3, 64, 64)
    with torch.no_grad():
        y = can(x.to(dev))
        acts = can.canonicalization_info_dict["group_activations"].cpu()
        gidx = can.canonicalization_info_dict["group_index"].cpu().long()
        loss = can.get_optimization_specific_loss().cpu()
        # oracle
        xin = io.pre_canonicalization_transform(x, (3, 64, 64), 0.8, 32)
        orbit = io.orbit_expand(xin, N, group_type, 32)
        from oracle import nets as onets
        vec = onets.conv_network(orbit, sd_net, 2, training=False)      # the oracle restatement of ConvNetwork (pinned to the
        acts_ref = io.optimized_group_activations(vec, ref_vec, G)      # reference's own class by tests/golden/conv_network.pt)
        loss_ref = io.optimization_specific_loss(vec, G, 16)
    assert acts.shape == (6, G)
    assert torch.allclose(acts, acts_ref, atol=2e-4, rtol=1e-3)
    top2 = acts_ref.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3
    assert torch.equal(gidx[clear], acts_ref.argmax(-1)[clear])
    assert torch.allclose(loss, loss_ref, atol=1e-4, rtol=1e-3)
    el = io.group_element_from_activations(acts, N, group_type, 1.0, training=False)
    _close(y, io.canonicalize_images(x, el["rotation"], el.get("reflection"), (3, 64, 64)))
    assert can.reference_vector.requires_grad is False and "reference_vector" in can.state_dict()
    # artifact branch (random rotate-and-back through the generic group-action entry point) runs and is finite
    hp2 = types.SimpleNamespace(**{**vars(hp), "artifact_err_wt": 1.0})
    can2 = ea.OptimizedGroupEquivariantImageCanonicalization(copy.deepcopy(cpu_net), hp2, (3, 64, 64)).to(dev).eval()
    with torch.no_grad():
        can2(x.to(dev))
        assert torch.isfinite(can2.get_optimization_specific_loss()).item()
    assert "vector_out_dummy" in can2.canonicalization_info_dict


def test_cfg5_shape_d4_1024(dev):
    """BASELINE config 5 shape: 1024x1024 D4 (one image per element), canonicalize + scalar invert + mask action."""
    from equiadapt_amd import ops
    from equiadapt_amd.images.utils import canonicalize_masks, device_tables

    torch.manual_seed(18)
    x = torch.nn.functional.avg_pool2d(torch.randn(8, 3, 1028, 1028), 5, 1, 0)  # smooth-ish, 1024x1024
    gidx = torch.arange(8)
    ang = torch.cat([io.group_angles(4)] * 2)[gidx]
    ref = (gidx >= 4).float()
    th, fl = device_tables("canonicalize", 4, True, (2048, 2048), dev)
    y = ops.canon_transform(x.to(dev), gidx.to(dev, torch.int32), th, fl, 512).cpu()
    # D4 = multiples of 90 degrees: exact arithmetic is a permutation of the pixels
    for e in range(8):
        img = x[e].flip(-1) if e >= 4 else x[e]
        want = torch.rot90(img, -(e % 4), (-2, -1))
        assert (y[e] - want).abs().max().item() <= 2e-4, e
    thi, fli, _ = device_tables("invert", 4, True, (1024, 1024), dev)
    back = ops.invert_action(y.to(dev), gidx.to(dev, torch.int32), thi, fli, None).cpu()
    # reference convention: invert flips when the indicator is 0 -> for D4 this is NOT the inverse of canonicalize
    # (SURVEY 8a I7 note); check against the oracle's op sequence on one sample instead of assuming a round trip
    w = io.invert_action(y[5:6], ang[5:6], ref[5:6], 4, 8, "scalar")
    assert (back[5:6] - w).abs().max().item() <= 1e-3
    masks = [(torch.rand(1, 1024, 1024) > 0.5).to(torch.uint8).to(dev) for _ in range(8)]
    out = canonicalize_masks(masks, gidx.to(dev, torch.int32), 4, flip_all=True)
    for e in (1, 6):
        assert torch.equal(out[e].cpu(), io.rotate_masks(io.flip_masks(masks[e].cpu()), -ang[e].item()))


def test_channels_last_kernels(dev):
    """eqa_window_sums_nhwc / eqa_bias_relu_nhwc against the NCHW kernel and plain torch."""
    from equiadapt_amd import ops

    torch.manual_seed(19)
    for (B, C, H, W, k) in [(3, 8, 20, 24, 5), (2, 256, 88, 88, 5), (2, 12, 11, 13, 3), (1, 4, 9, 9, 1), (2, 68, 30, 30, 5)]:
        x = torch.randn(B, C, H, W, device=dev)
        scale, shift = (torch.rand(C, device=dev) + 0.5), torch.randn(C, device=dev) * 0.3
        xcl = x.contiguous(memory_format=torch.channels_last)
        for relu in (True, False):
            want = ops.window_sums(x, k, scale, shift, relu)
            got = ops.window_sums(xcl, k, scale, shift, relu)
            assert torch.allclose(got, want, atol=2e-3, rtol=1e-6), (B, C, H, W, k, relu)  # fp32 segment partials
        bias = torch.randn(C, device=dev)
        y = xcl.clone()
        ops.bias_relu_nhwc_(y, bias)
        assert torch.equal(y, torch.relu(x + bias[None, :, None, None]))


def test_step_is_hipgraph_capturable(dev):
    """canonicalize + invert launch only stream-ordered kernels and never sync the host, so a whole inference step can
    be captured into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed on new data."""
    import equiadapt_amd as ea

    torch.manual_seed(20)
    net = ea.CustomEquivariantNetwork((3, 32, 32), 4, 5, "rotation", 8, 2, device="cpu")
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=0.8, resize_shape=32)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 64, 64)).to(dev).eval()
    x_static = torch.randn(8, 3, 64, 64, device=dev)
    f_static = torch.randn(8, 3, 64, 64, device=dev)
    with torch.no_grad():
        for _ in range(2):  # warm-up outside capture (table uploads, MIOpen search)
            can(x_static)
            can.invert_canonicalization(f_static, induced_rep_type="scalar")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y_static = can(x_static)
            inv_static = can.invert_canonicalization(f_static, induced_rep_type="scalar")
            idx_static = can.canonicalization_info_dict["group_index"]
        for seed in (1, 2):
            xn = torch.randn(8, 3, 64, 64, generator=torch.Generator().manual_seed(seed)).to(dev)
            fn = torch.randn(8, 3, 64, 64, generator=torch.Generator().manual_seed(10 + seed)).to(dev)
            x_static.copy_(xn)
            f_static.copy_(fn)
            g.replay()
            torch.cuda.synchronize()
            y_g, inv_g, idx_g = y_static.clone(), inv_static.clone(), idx_static.clone()
            y_e = can(xn)
            inv_e = can.invert_canonicalization(fn, induced_rep_type="scalar")
            assert torch.equal(idx_g, can.canonicalization_info_dict["group_index"])
            assert torch.equal(y_g, y_e) and torch.equal(inv_g, inv_e)


@pytest.mark.parametrize("group_type,N", [("rotation", 4), ("roto-reflection", 4), ("rotation", 8)])
def test_group_inference_orbit_and_metrics(dev, group_type, N):
    """GroupInference (reference examples/images/classification/inference_utils.py): orbit bit-exact vs the oracle
    (nearest sampling moves pixels, it does not interpolate), metrics computed like the reference."""
    import equiadapt_amd as ea
    from equiadapt_amd.inference import get_inference_method

    torch.manual_seed(21)
    x = torch.randn(5, 3, 32, 40)
    hp = types.SimpleNamespace(method="group", group_type=group_type, num_rotations=N)
    net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(3 * 32 * 40, 4)).to(dev)
    inf = get_inference_method(ea.IdentityCanonicalization(), net, 4, hp, (3, 32, 40))
    orbit = inf.group_orbit(x.to(dev)).cpu()
    want = io.group_inference_orbit(x, N, group_type)
    assert orbit.shape == want.shape
    assert torch.equal(orbit, want)
    y = torch.tensor([0, 1, 2, 3, 0], device=dev)
    with torch.no_grad():
        m = inf.get_inference_metrics(x.to(dev), y)
        logits0 = net(want[0].to(dev))
    E = N if group_type == "rotation" else 2 * N
    assert set(f"test/acc_group_element_{i}" for i in range(E)) <= set(m)
    assert torch.isclose(m["test/acc"].cpu(), (logits0.argmax(-1) == y).float().mean().cpu())
    van = get_inference_method(ea.IdentityCanonicalization(), net, 4, types.SimpleNamespace(method="vanilla"))
    assert "test/acc" in van.get_inference_metrics(x.to(dev), y)
    with pytest.raises(ValueError):
        get_inference_method(None, None, 4, types.SimpleNamespace(method="ensemble"))


def test_winograd_conv_matches_direct(dev):
    """Winograd F(2x2,5x5) and F(4x4,5x5) (transform kernels + batched GEMM) vs F.conv2d in fp64.  Tolerance: 2e-5 of
    max|y| (measured ~7e-6 / ~9e-6 at 256 channels; MIOpen's direct fp32 conv: ~3e-7)."""
    import torch.nn.functional as F

    from equiadapt_amd import ops
    from equiadapt_amd.images.canonicalization_networks import winograd

    torch.manual_seed(22)
    seen = set()
    for (B, Cin, Cout, H, W) in [(3, 32, 48, 12, 16), (2, 64, 64, 30, 30), (70, 32, 32, 10, 10), (2, 256, 256, 20, 20),
                                 (300, 32, 32, 12, 12), (2, 64, 32, 24, 32)]:
        x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        g = torch.randn(Cout, Cin, 5, 5, device=dev) / (5 * Cin ** 0.5)
        bias = torch.randn(Cout, device=dev)
        assert winograd.applicable(x, Cin, Cout)
        exact = torch.relu(F.conv2d(x.double(), g.double(), bias.double()))
        scale = exact.abs().max().item()
        tiles = [2, 4] if winograd.tile_for(x) == 4 else [2]
        for m in tiles:
            seen.add(m)
            U = winograd.transform_filters(g, m)
            assert U.shape == ((m + 4) ** 2, Cin, Cout)
            got = winograd.conv5x5(x, U, bias, relu=True)
            assert got.is_contiguous(memory_format=torch.channels_last) and got.shape == (B, Cout, H - 4, W - 4)
            e_w = (got.double() - exact).abs().max().item()
            assert e_w <= 2e-5 * scale, (m, e_w, scale)
            got2 = winograd.conv5x5(x, U, None, relu=False)
            assert (got2.double() - F.conv2d(x.double(), g.double())).abs().max().item() <= 2e-5 * scale
            # fused tail: window sums of the activation straight from the output transform
            if winograd.sums_applicable(x, 5, m):
                S = winograd.conv5x5(x, U, bias, relu=True, sums_k=5)
                S_ref = ops.window_sums(got, 5)
                assert torch.allclose(S, S_ref, atol=1e-3, rtol=1e-6), (m, B, Cin, Cout, H, W)
            if m == 2 and winograd.sums_applicable(x, 3, m):
                S = winograd.conv5x5(x, U, bias, relu=True, sums_k=3)
                assert torch.allclose(S, ops.window_sums(got, 3), atol=1e-3, rtol=1e-6)
            # fused input activation: conv(relu(x + in_bias))
            ib = torch.randn(Cin, device=dev)
            got3 = winograd.conv5x5(x, U, bias, relu=True, in_bias=ib, in_relu=True)
            want3 = torch.relu(F.conv2d(torch.relu(x.double() + ib.double()[None, :, None, None]), g.double(), bias.double()))
            assert (got3.double() - want3).abs().max().item() <= 2e-5 * want3.abs().max().item()
            # zero padding inside the input transform (no padded copy): against the fp64 convolution with padding=4, same bound
            if (H + 8 - 4) % m == 0 and (W + 8 - 4) % m == 0:
                a = winograd.conv5x5(x, U, None, relu=False, pad=4)
                want_p = F.conv2d(x.double(), g.double(), padding=4)
                assert a.shape == want_p.shape == (B, Cout, H + 4, W + 4)
                assert (a.double() - want_p).abs().max().item() <= 2e-5 * want_p.abs().max().item(), (m, H, W)
    assert seen == {2, 4}
    # sizes the large tile does not divide are refused by the entry point itself
    from equiadapt_amd import _lib
    lib = _lib.load()
    x = torch.randn(1, 10, 10, 32, device=dev)
    V = torch.empty(64 * 32 * 4, device=dev)
    assert lib.eqa_winograd_f4k5_input(x.data_ptr(), V.data_ptr(), None, 0, 1, 10, 10, 32, None) == -3  # EQA_ERR_UNSUPPORTED


def test_nbody_e3_canonicalizer_matches_reference_golden(dev, golden):
    """(f).4: EuclideanGroupNBody canonicalize / invert on the HIP kernels vs reference-generated vectors."""
    import equiadapt_amd as ea

    g = golden("nbody.pt")
    rot_vec, trans = g["rot_vec"].to(dev), g["trans"].to(dev)

    class FakeNet(torch.nn.Module):
        def forward(self, nodes, loc, edges, vel, edge_attr, charges):
            return rot_vec, trans

    can = ea.EuclideanGroupNBody(FakeNet())
    with torch.no_grad():
        cl, cv = can(torch.zeros(40, 1, device=dev), loc=g["loc"].to(dev), edges=None, vel=g["vel"].to(dev), edge_attr=None, charges=None)
        inv = can.invert_canonicalization(g["pred"].to(dev))
    R = can.canonicalization_info_dict["group_element"]["rotation_matrix"].cpu()
    assert torch.allclose(R, g["rotation"], atol=1e-5)
    # coordinates: |x| up to ~3 times the 1e-5 allowed on R (a few of the 40 random frames are nearly collinear)
    assert torch.allclose(cl.cpu(), g["canonical_loc"], atol=5e-5) and torch.allclose(cv.cpu(), g["canonical_vel"], atol=5e-5)
    assert torch.allclose(inv.cpu(), g["inverted"], atol=5e-5)
    # autograd path (op-by-op) gives the same numbers
    rv = rot_vec.clone().requires_grad_(True)
    assert torch.allclose(can.modified_gram_schmidt(rv).detach().cpu(), g["rotation"], atol=1e-5)


def test_lift_conv_mfma_matches_conv2d(dev):
    """eqa_lift_conv_nhwc (fp32 MFMA implicit GEMM) vs F.conv2d in fp64.  Tolerance 2e-6 of max|y|: the MFMA is an exact
    fp32 fmaf chain over K <= 80 terms, same class of rounding as the direct fp32 convolution."""
    import torch.nn.functional as F

    from equiadapt_amd import _lib, ops

    torch.manual_seed(31)
    for (B, Cin, K, Cout, H, W) in [(3, 3, 5, 256, 20, 24), (2, 3, 3, 64, 9, 11), (5, 2, 5, 128, 13, 13), (1, 4, 3, 64, 40, 7),
                                    (2, 5, 3, 192, 12, 12), (64, 3, 5, 64, 33, 33), (1, 3, 5, 64, 5, 5), (7, 3, 5, 128, 6, 70),
                                    (300, 3, 3, 64, 10, 37),
                                    # channel counts off the 64-multiples (predicated stores, several tiles per row)
                                    (128, 3, 5, 32, 32, 32), (3, 3, 5, 16, 20, 75), (2, 3, 3, 48, 40, 40), (4, 3, 5, 80, 12, 100),
                                    (1, 3, 5, 32, 5, 5),   # incl. streams of 1, 2, 3 tiles per wave and partial tiles
                                    # the dense form (K = 76, tiles over the flattened map): one tile, tiles that run over a row
                                    # end at every position, an overlapping last tile, the headline shape, many images
                                    (1, 3, 5, 64, 5, 36), (3, 3, 5, 128, 9, 37), (2, 3, 5, 64, 96, 96), (2, 3, 5, 256, 37, 67),
                                    (300, 3, 5, 64, 8, 40), (1, 3, 5, 64, 6, 36)]:
        assert ops.lift_conv_supported(Cin, K, K, Cout)
        x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(Cout, Cin, K, K, device=dev) / (K * Cin ** 0.5)
        bias = torch.randn(Cout, device=dev)
        wpk = ops.pack_lift_weights(w)
        for (b, relu) in [(bias, True), (None, False), (bias, False)]:
            got = ops.lift_conv_nhwc(x, wpk, b, relu, K, K)
            assert got.shape == (B, Cout, H - K + 1, W - K + 1) and got.is_contiguous(memory_format=torch.channels_last)
            want = F.conv2d(x.double(), w.double(), None if b is None else b.double())
            want = torch.relu(want) if relu else want
            scale = want.abs().max().item()
            assert (got.double() - want).abs().max().item() <= 2e-6 * scale, (B, Cin, K, Cout, H, W)
    # input at the very end of an allocation: the last receptive field must not be over-read (exact-size buffer)
    x = torch.randn(1, 3, 5, 5, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 3, 5, 5, device=dev)
    got = ops.lift_conv_nhwc(x, ops.pack_lift_weights(w), None, False, 5, 5)
    assert torch.allclose(got.double(), F.conv2d(x.double(), w.double()), atol=1e-4)
    for hw in [(5, 36), (6, 37), (7, 41)]:   # ... and the dense form's longer segments (exact-size buffers)
        x = torch.randn(1, 3, *hw, device=dev).contiguous(memory_format=torch.channels_last)
        got = ops.lift_conv_nhwc(x, ops.pack_lift_weights(w), None, False, 5, 5)
        assert torch.allclose(got.double(), F.conv2d(x.double(), w.double()), atol=1e-4), hw
    x = torch.randn(2, 3, 9, 40, device=dev).contiguous(memory_format=torch.channels_last)
    x[1, 2, 3, 37] = float("inf")            # confined to its receptive fields in the dense form too (row-crossing tiles)
    got = ops.lift_conv_nhwc(x, ops.pack_lift_weights(w), None, False, 5, 5)
    bad = ~torch.isfinite(got).all(dim=1)
    want_bad = torch.zeros_like(bad)
    want_bad[1, 0:4, 33:36] = True
    assert torch.equal(bad, want_bad)
    # non-finite pixels stay confined to the outputs whose receptive field contains them (zero-weight duplicates are
    # always elements of the same receptive field)
    x = torch.randn(1, 3, 12, 12, device=dev).contiguous(memory_format=torch.channels_last)
    x[0, 1, 11, 11] = float("inf")
    got = ops.lift_conv_nhwc(x, ops.pack_lift_weights(w), None, False, 5, 5)
    assert torch.isfinite(got[0, :, :7, :]).all() and torch.isfinite(got[0, :, :, :7]).all() and not torch.isfinite(got[0, :, 7, 7]).all()
    # unsupported shapes are refused, not approximated
    lib = _lib.load()
    assert not ops.lift_conv_supported(1, 5, 5, 64) and not ops.lift_conv_supported(3, 5, 5, 24) and not ops.lift_conv_supported(3, 7, 7, 64)
    assert not ops.lift_conv_supported(4, 3, 4, 64)      # R = 16: no spare k-slot for the bias
    assert lib.eqa_lift_conv_nhwc(x.data_ptr(), w.data_ptr(), None, 0, got.data_ptr(), 1, 12, 12, 1, 5, 5, 64, None) == -3
    assert lib.eqa_lift_conv_nhwc(x.data_ptr(), w.data_ptr(), None, 0, got.data_ptr(), 1, 12, 12, 4, 3, 4, 64, None) == -3


def test_window_sums_gemv_matches_matmul(dev):
    """eqa_window_sums_gemv vs the fp64 matmul it replaces (fp32 output: equal to within one rounding)."""
    from equiadapt_amd import _lib, ops

    torch.manual_seed(41)
    for (B, K, E) in [(5, 6400, 8), (1, 75, 1), (257, 1000, 16), (3, 256, 4)]:
        S = torch.randn(B, K, dtype=torch.float64, device=dev) * 100
        Wm = torch.randn(E, K, dtype=torch.float64, device=dev)
        want = (S @ Wm.t() * 0.37 + 1.25)
        got = ops.window_sums_gemv(S, Wm, 0.37, 1.25)
        assert got.dtype == torch.float32 and got.shape == (B, E)
        assert torch.allclose(got.double(), want, rtol=2e-7, atol=1e-9 * want.abs().max().item())
        got_t = ops.window_sums_gemv(S, Wm, 0.37, torch.tensor(1.25, dtype=torch.float64, device=dev))
        assert torch.allclose(got_t.double(), want, rtol=4e-7, atol=1e-6)
    lib = _lib.load()
    assert lib.eqa_window_sums_gemv(S.data_ptr(), Wm.data_ptr(), got.data_ptr(), 3, 256, 17, 1.0, 0.0, None) == -3
    assert lib.eqa_window_sums_gemv(None, None, None, 0, 256, 4, 1.0, 0.0, None) == 0


class _FixedVectorNet(torch.nn.Module):
    """Stands in for a steerable network: returns its (B, n_vectors, 2) parameter whatever the input."""

    group_type = "rotation"

    def __init__(self, vectors):
        super().__init__()
        self.vectors = torch.nn.Parameter(vectors)

    def forward(self, x):
        return self.vectors[: x.shape[0]]


@pytest.mark.parametrize("shape", [(3, 32, 32), (3, 40, 56), (1, 28, 28)])
def test_steerable_canonicalizer_matches_oracle(dev, shape):
    """(f).4 SteerableImageCanonicalization: forward vs the oracle's pad -> warp_affine -> crop, the mutated info dict,
    and the gradients w.r.t. the network output and the image vs autograd through the oracle."""
    import types

    import equiadapt_amd as ea
    from oracle import image_ops as o

    torch.manual_seed(51)
    B = 5
    C, H, W = shape
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    x = torch.stack([torch.stack([torch.sin(3 * xx + b + c) * torch.cos(2 * yy - c) + 0.3 * xx * yy for c in range(C)]) for b in range(B)])
    x = x + 0.05 * torch.randn(B, C, H, W)
    vec = torch.randn(B, 1, 2)
    hp = types.SimpleNamespace(input_crop_ratio=0.9, resize_shape=(16, 16))
    can = ea.SteerableImageCanonicalization(_FixedVectorNet(vec.clone().to(dev)), hp, shape).to(dev)
    xd = x.to(dev).requires_grad_(True)
    got = can.canonicalize(xd)
    # oracle (CPU autograd)
    xo = x.clone().requires_grad_(True)
    vo = vec.clone().requires_grad_(True)
    want, R_after = o.canonicalize_images_continuous(xo, o.steerable_rotation_from_vector(vo[:, 0]), gray=(C == 1))
    assert got.shape == want.shape == (B, C, H, W)
    assert (got.detach().cpu() - want.detach()).abs().max().item() <= PIX_MAX
    info = can.canonicalization_info_dict
    assert torch.allclose(info["group_element"]["rotation"].detach().cpu(), R_after.detach(), atol=1e-6)
    assert info["group_element_matrix_representation"] is info["group_element"]["rotation"]
    from oracle import pointcloud_ops as po

    assert torch.allclose(can.get_prior_regularization_loss().detach().cpu(), po.continuous_prior_loss(R_after.detach()), atol=1e-6)
    assert torch.allclose(can.get_identity_metric().detach().cpu(), po.continuous_identity_metric(R_after.detach()), atol=1e-6)
    wgt = torch.randn(B, C, H, W)
    (got * wgt.to(dev)).sum().backward()
    (want * wgt).sum().backward()
    gv, gv_o = can.canonicalization_network.vectors.grad.cpu(), vo.grad
    scale = gv_o.abs().max().item()
    assert (gv - gv_o).abs().max().item() <= 5e-3 * scale, (gv, gv_o)
    assert (xd.grad.cpu() - xo.grad).abs().max().item() <= 1e-3 * max(xo.grad.abs().max().item(), 1.0)
    # reference error behaviour
    with pytest.raises(KeyError):
        can.invert_canonicalization(got.detach())
    can.group_type = "roto-reflection"
    with pytest.raises(NotImplementedError):
        can.canonicalize(x.to(dev))


def test_optimized_steerable_canonicalizer(dev):
    """(f).4 OptimizedSteerableImageCanonicalization: group_augment vs the oracle (including the reference's (4,B)->(B,2,2)
    reshape that mixes samples), the regression loss, and one optimisation step end to end."""
    import types

    import equiadapt_amd as ea
    from oracle import image_ops as o

    torch.manual_seed(52)
    B, C, H, W = 4, 3, 32, 32
    x = torch.randn(B, C, H, W)
    hp = types.SimpleNamespace(input_crop_ratio=0.9, resize_shape=(16, 16), group_type="rotation")
    net = ea.ConvNetwork((3, 16, 16), out_channels=8, kernel_size=3, num_layers=2, out_vector_size=2).to(dev)
    can = ea.OptimizedSteerableImageCanonicalization(net, hp, (C, H, W)).to(dev)
    angles = torch.tensor([0.3, 1.1, 2.5, 4.0])
    aug, gt = can.group_augment(x.to(dev), angles.to(dev))
    aug_o, gt_o = o.continuous_group_augment(x, angles)
    assert torch.allclose(gt.cpu(), gt_o, atol=1e-6)
    assert (aug.cpu() - aug_o).abs().max().item() <= PIX_MAX
    assert not torch.allclose(gt_o[0], torch.tensor([[math.cos(0.3), math.sin(0.3)], [-math.sin(0.3), math.cos(0.3)]]), atol=1e-3)  # the quirk
    # single image: a plain rotation
    aug1, gt1 = can.group_augment(x[:1].to(dev), angles[:1].to(dev))
    aug1_o, gt1_o = o.continuous_group_augment(x[:1], angles[:1])
    assert torch.allclose(gt1.cpu(), gt1_o, atol=1e-6) and (aug1.cpu() - aug1_o).abs().max().item() <= PIX_MAX
    # end to end: forward, losses, backward
    can.train()
    opt = torch.optim.SGD(can.parameters(), lr=1e-2)
    y = can(x.to(dev))
    assert y.shape == (B, C, H, W) and torch.isfinite(y).all()
    info = can.canonicalization_info_dict
    want_loss = o.continuous_optimization_loss(info["group_element_matrix_representation_augmented"].detach().cpu(),
                                               info["group_element_matrix_representation_augmented_gt"].cpu())
    loss = can.get_optimization_specific_loss()
    assert torch.allclose(loss.detach().cpu(), want_loss, atol=1e-6)
    total = loss + can.get_prior_regularization_loss() + y.square().mean()
    total.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in can.parameters())
    opt.step()
    assert 0.0 <= float(1.0 - can.get_identity_metric().detach()) < 4.0


def test_nearest_action_tiled_equals_direct(dev):
    """The LDS-tiled nearest-neighbour kernel (default) and the row-per-block kernel (eqa_set_option(0, 1)) give identical
    bytes / floats: odd sizes, 45-degree elements, flips, padded frames with a cropped window, plane reuse (src_mod)."""
    from equiadapt_amd import _lib, ops
    from equiadapt_amd.images import geometry

    lib = _lib.load()
    torch.manual_seed(81)
    for (H, W, N) in [(70, 70, 8), (129, 65, 4), (1024, 1024, 4), (33, 200, 8)]:
        masks = (torch.rand(11, H, W, device=dev) > 0.5).to(torch.uint8) * torch.randint(1, 255, (11, 1, 1), device=dev, dtype=torch.uint8)
        ang = torch.cat([geometry.group_angles(N)] * 2)
        rtheta = geometry.mask_rotation_table(-ang, (H, W)).to(dev)
        flags = torch.cat([torch.zeros(N), torch.ones(N)]).to(dev, torch.int32)
        eidx = torch.randint(0, 2 * N, (11,), device=dev, dtype=torch.int32)
        a = ops.mask_action_nearest(masks, eidx, rtheta, flags)
        lib.eqa_set_option(0, 1)
        try:
            b = ops.mask_action_nearest(masks, eidx, rtheta, flags)
        finally:
            lib.eqa_set_option(0, 0)
        assert torch.equal(a, b), (H, W, N)
    # fp32 images, padded frame, cropped window, each source plane used by several outputs
    x = torch.randn(6, 50, 50, device=dev)
    pad, OH, OW, top, left = 20, 50, 50, 20, 20
    ang = geometry.group_angles(8)
    rtheta = geometry.mask_rotation_table(ang, (50 + 2 * pad, 50 + 2 * pad)).to(dev)
    eidx = (torch.arange(48, device=dev) // 6).to(torch.int32)
    a = ops.image_action_nearest(x, eidx, rtheta, None, pad, (OH, OW), (top, left), 48, 6)
    lib.eqa_set_option(0, 1)
    try:
        b = ops.image_action_nearest(x, eidx, rtheta, None, pad, (OH, OW), (top, left), 48, 6)
    finally:
        lib.eqa_set_option(0, 0)
    assert torch.equal(a, b)


@pytest.mark.parametrize("group_type,N", [("rotation", 8), ("roto-reflection", 4)])
def test_escnn_four_layers_two_winograd_layers(dev, group_type, N, monkeypatch):
    """ESCNNEquivariantNetwork with num_layers = 4 and 64 channels (8 fields x |G| = 8): lifting MFMA conv -> two Winograd
    F(4x4,5x5) layers in sequence (the second consumes the first through the fused bias/ReLU, the last one emits window sums)
    -> linear tail.  Inference against the oracle; training path against the plain module sequence."""
    import copy

    import equiadapt_amd as ea
    from oracle import nets as onets

    torch.manual_seed(111)
    net = ea.ESCNNEquivariantNetwork((3, 36, 36), 8, 5, group_type, N, 4)
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.normal_(0.1, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net = net.to(dev).eval()
    x = torch.randn(6, 3, 36, 36)
    with torch.no_grad():
        fast = net(x.to(dev)).cpu()
    G = N if group_type == "rotation" else 2 * N
    assert fast.shape == (6, G)
    want = onets.escnn_like_network(x, {k: v.cpu() for k, v in net.state_dict().items()}, group_type, N, 4, 8)
    assert torch.allclose(fast, want, atol=5e-5, rtol=1e-3), (fast - want).abs().max().item()
    # training
    ref = copy.deepcopy(net)
    net.train()
    ref.train()
    xd = x.to(dev)
    w = torch.randn(6, G, device=dev)
    a1 = net(xd)
    monkeypatch.setenv("EQA_TRAIN_FAST", "0")
    a2 = ref(xd)
    monkeypatch.delenv("EQA_TRAIN_FAST")
    assert (a1 - a2).abs().max().item() <= 5e-5 * max(a2.abs().max().item(), 1.0)
    (a1 * w).sum().backward()
    (a2 * w).sum().backward()
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), ref.named_parameters()):
        g = p2.grad.abs().max().item()
        if g <= 1e-5:
            continue
        # (batch statistics over 6 images amplify the Winograd transforms' 1e-5 rounding; the plane GEMM's own summation order --
        # exact fmaf chains, 1e-6 against fp64 in test_plane_gemm_matches_the_batched_product -- puts the worst entry at 0.84 %; the bound was 0.5 % with the library's GEMM)
        assert (p1.grad - p2.grad).abs().max().item() <= 1e-2 * g, (n1, (p1.grad - p2.grad).abs().max().item(), g)


def test_boxes_action_matches_flip_and_rotate_boxes(dev):
    """eqa_boxes_action (one launch for every box of the batch) vs the reference's op sequence flip_boxes -> rotate_boxes
    per sample (images/utils.py:97-109,161-187), incl. the in-place flip of the caller's tensors and empty box lists.
    Same fp32 arithmetic; allowed difference: 2 ulp of the coordinate range (cos / sin come from the same library)."""
    from equiadapt_amd.images.utils import canonicalize_boxes, flip_boxes, rotate_boxes

    torch.manual_seed(77)
    W = 1024
    counts = [3, 0, 7, 1, 12, 5]
    rot = torch.tensor([0.0, 45.0, 90.0, 135.0, 270.0, 315.0], device=dev)
    for flip_all in (False, True):
        boxes = []
        for n in counts:
            xy = torch.rand(n, 2, 2, device=dev) * W
            boxes.append(torch.cat([xy.min(1).values, xy.max(1).values], dim=1).contiguous())
        mine_in = [b.clone() for b in boxes]
        ref_in = [b.clone() for b in boxes]
        want = []
        for t, b in enumerate(ref_in):
            if flip_all:
                b = flip_boxes(b, W)
            want.append(rotate_boxes(b, rot[t], W))
        got = canonicalize_boxes(mine_in, rot, W, flip_all)
        for t in range(len(counts)):
            assert got[t].shape == (counts[t], 4)
            assert torch.equal(mine_in[t], ref_in[t])                      # the caller's tensors: flipped in place or untouched
            if counts[t]:
                assert (got[t] - want[t]).abs().max().item() <= 2.5e-4, (flip_all, t, (got[t] - want[t]).abs().max().item())
                assert (got[t][:, 0] <= got[t][:, 2]).all() and (got[t][:, 1] <= got[t][:, 3]).all()


def test_fft48_convolution_matches_conv2d(dev):
    """eqa_fft48k5_* (overlap-save FFT convolution, 48x48 tiles, complex GEMM as a real one) vs F.conv2d in fp64: plain
    output and the fused window sums, tiles that fit exactly (92 -> 88 = 2 x 44), partial tiles, one tile, non-square maps,
    the previous layer's bias + ReLU on the loads.  Tolerance 5e-6 of max|y| (measured 2-4e-7; Winograd F(4,5): 9e-6)."""
    import torch.nn.functional as F

    from equiadapt_amd import _lib
    from equiadapt_amd.images.canonicalization_networks import fftconv

    torch.manual_seed(48)
    cases = [(2, 8, 12, 92, 92), (3, 16, 8, 60, 97), (1, 4, 4, 48, 48), (2, 4, 8, 20, 33), (1, 64, 64, 92, 92), (2, 12, 4, 137, 49),
             # channel counts that are multiples of 16 take the fused kernels: ragged last tiles, the right border columns
             # of the window sums inside one tile (97), split over two (49, 50, 51) and alone in the last tile (52)
             (2, 8, 16, 137, 49), (1, 16, 32, 60, 97), (1, 8, 16, 53, 50), (1, 16, 16, 48, 51), (1, 8, 16, 50, 52), (1, 8, 16, 30, 140),
             # 100 tiles x 3 channel groups = 300 work items: the persistent inverse pipeline's blocks take one or two items, and
             # the XCD split has a remainder (300 = 8 x 37 + 4)
             (25, 16, 48, 92, 92)]
    for (B, Cin, Cout, H, W) in cases:
        x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(Cout, Cin, 5, 5, device=dev) / (5 * Cin ** 0.5)
        b, ib = torch.randn(Cout, device=dev), torch.randn(Cin, device=dev)
        Bm = fftconv.filter_spectra(w)
        assert Bm.shape == (fftconv.F, 2 * Cin, 2 * Cout) and fftconv.F == 1154
        for (relu, in_relu) in [(False, False), (True, True)]:
            xin = torch.relu(x.double() + ib.double()[None, :, None, None]) if in_relu else x.double()
            want = F.conv2d(xin, w.double(), b.double())
            want = torch.relu(want) if relu else want
            got = fftconv.conv5x5(x, Bm, b, relu, ib if in_relu else None, in_relu)
            assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
            scale = want.abs().max().item()
            assert (got.double() - want).abs().max().item() <= 5e-6 * scale, (B, Cin, Cout, H, W, relu)
            OH, OW = H - 4, W - 4
            for k in (5, 3):
                if min(OH, OW) < 2 * k - 1:
                    continue
                S = fftconv.conv5x5(x, Bm, b, relu, ib if in_relu else None, in_relu, sums_k=k)
                Sw = torch.stack([torch.stack([want[:, :, u:u + OH - k + 1, v:v + OW - k + 1].sum((-1, -2)) for v in range(k)], -1)
                                  for u in range(k)], -2)
                assert S.dtype == torch.float64 and ((S - Sw).abs().max() <= 2e-6 * Sw.abs().max().clamp_min(scale)), (B, H, W, k)
    # which shapes take this path: tiles must fit the output to within 12 %
    xs = torch.zeros(8, 4, 92, 92, device=dev).contiguous(memory_format=torch.channels_last)
    assert fftconv.applicable(xs, 4, 4) and fftconv.tiles(92) == 2 and fftconv.tiles(93) == 3
    assert not fftconv.applicable(xs[:4], 4, 4)            # too few tiles to amortise streaming the filter spectra
    assert not fftconv.applicable(torch.zeros(8, 4, 60, 60, device=dev).contiguous(memory_format=torch.channels_last), 4, 4)
    assert _lib.load().eqa_fft48k5_tiles(92) == 2


def test_escnn_inference_fft_path_equals_winograd_and_module_paths(dev, monkeypatch):
    """ESCNNEquivariantNetwork at the headline geometry (96 -> lift 92 -> 5x5 -> 88 -> linearised tail), reduced width:
    the FFT path (default), the Winograd path and the plain module sequence give the same activations and group index."""
    import equiadapt_amd as ea
    from equiadapt_amd.images.canonicalization_networks import fftconv

    torch.manual_seed(9)
    net = ea.ESCNNEquivariantNetwork((3, 96, 96), out_channels=8, kernel_size=5, group_type="rotation", num_rotations=8,
                                     num_layers=3).to(dev).eval()
    for m in net.modules():  # non-trivial batch-norm statistics
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(9, 3, 96, 96, device=dev)
    with torch.no_grad():
        assert fftconv.ENABLED and fftconv.applicable(torch.zeros(9, 64, 92, 92, device=dev).contiguous(memory_format=torch.channels_last), 64, 64)
        a_fft = net(x)
        monkeypatch.setattr(fftconv, "ENABLED", False)
        a_wino = net(x)
        a_mod = ea.images.canonicalization_networks.pooling.group_pool(net.eqv_network(x))
    scale = a_mod.abs().max().item()
    assert (a_fft - a_mod).abs().max().item() <= 2e-5 * max(scale, 1.0)
    assert (a_wino - a_mod).abs().max().item() <= 2e-5 * max(scale, 1.0)
    assert torch.equal(a_fft.argmax(1), a_mod.argmax(1))


def test_fft_filter_spectra_kernel_matches_host_construction(dev):
    """eqa_fft48k5_filter_spectra (one kernel, fp64 accumulation) vs the host construction through torch.fft in fp64, for the
    grouped ([Re x 16 | Im x 16]) and the interleaved row order."""
    from equiadapt_amd.images.canonicalization_networks import fftconv

    torch.manual_seed(12)
    for (Cout, Cin) in [(64, 32), (24, 12), (256, 256), (5, 3)]:
        w = torch.randn(Cout, Cin, 5, 5, device=dev)
        got = fftconv.filter_spectra(w)
        want = fftconv.filter_spectra(w, groups=fftconv.group_sizes(Cin, Cout))      # host path (explicit groups)
        assert got.shape == want.shape == (fftconv.F, 2 * Cin, 2 * Cout)
        assert (got - want).abs().max().item() <= 2e-7 * want.abs().max().item() + 1e-12, (Cout, Cin)


@pytest.mark.parametrize("group_type,N", [("rotation", 8), ("roto-reflection", 4)])
def test_headline_geometry_end_to_end_against_oracle_fft_path(dev, group_type, N):
    """The headline pipeline at its real geometry -- 224x224x3, crop 0.8, resize 96, lift 5x5 -> 92, 5x5 -> 88, linearised tail --
    with 8 fields (64 channels) and 9 images, so that the hidden layer takes the FFT convolution (36 tiles).  Activations,
    group index, canonicalized images and the inverted prediction against the CPU oracle."""
    import equiadapt_amd as ea
    from equiadapt_amd.images.canonicalization_networks import fftconv
    from oracle import nets as onets

    torch.manual_seed(2024)
    net = ea.ESCNNEquivariantNetwork((3, 96, 96), 8, 5, group_type, N, 3)
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.normal_(0.1, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    with torch.no_grad():                     # a random-init net separates the orientations of noise by ~1e-5: widen the margins
        [m for m in net.eqv_network if hasattr(m, "expanded_weights")][-1].weights.mul_(100.0)
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=0.8, resize_shape=96)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 224, 224)).to(dev).eval()
    G = can.num_group
    assert fftconv.ENABLED and fftconv.applicable(
        torch.zeros(9, 8 * G, 92, 92, device=dev).contiguous(memory_format=torch.channels_last), 8 * G, 8 * G)
    x = torch.randn(9, 3, 224, 224)
    f = torch.randn(9, 3, 224, 224)
    with torch.no_grad():
        y = can(x.to(dev))
        acts = can.canonicalization_info_dict["group_activations"].cpu()
        gidx = can.canonicalization_info_dict["group_index"].cpu().long()
        inv = can.invert_canonicalization(f.to(dev), induced_rep_type="scalar")
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    xin = io.pre_canonicalization_transform(x, (3, 224, 224), 0.8, 96)
    acts_ref = onets.escnn_like_network(xin, sd, group_type, N, 3, 8)
    scale = acts_ref.abs().max().item()
    assert (acts - acts_ref).abs().max().item() <= 2e-5 * max(scale, 1.0), ((acts - acts_ref).abs().max().item(), scale)
    top2 = acts_ref.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4 * max(scale, 1.0)
    assert clear.sum() >= 5 and torch.equal(gidx[clear], acts_ref.argmax(-1)[clear]), clear.sum()
    el = io.group_element_from_activations(acts, N, group_type, 1.0, training=False)
    _close(y, io.canonicalize_images(x, el["rotation"], el.get("reflection"), (3, 224, 224)))      # white-noise images: the
    _close(inv, io.invert_action(f, el["rotation"], el.get("reflection"), N, G, "scalar"))           # loose pixel budget


@pytest.mark.parametrize("group_type,N", [("rotation", 8), ("roto-reflection", 4)])
def test_bench_configuration_full_width_against_oracle(dev, group_type, N):
    """The configuration the metric is quoted on, at its own width: exactly bench.build_canonicalizer (32 fields x 8
    orientations = 256 channels, k5, 3 layers, 224 -> crop 0.8 -> 96), 8 images so that the hidden layer runs the FFT
    convolution on 256 channels (32 tiles, odd tile pitch, fft48_inv_fused_kernel<4,16>), plus the D4 bank of the same width.
    Activations, group index, canonicalized images and invert(scalar) against the CPU oracle
    (reference: examples/images/classification/configs/canonicalization/group_equivariant.yaml:3-11,
    equiadapt/images/canonicalization_networks/escnn_networks.py:93-117)."""
    import bench
    from equiadapt_amd.images.canonicalization_networks import fftconv
    from oracle import nets as onets

    can = bench.build_canonicalizer(dev, group_type=group_type, num_rotations=N)
    net = can.canonicalization_network
    G = can.num_group
    assert G == 8 and net.out_channels == 32
    B = 8
    assert fftconv.ENABLED and fftconv.applicable(
        torch.zeros(B, 32 * G, 92, 92, device=dev).contiguous(memory_format=torch.channels_last), 32 * G, 32 * G)
    x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    f = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        y = can(x.to(dev))
        acts = can.canonicalization_info_dict["group_activations"].cpu()
        gidx = can.canonicalization_info_dict["group_index"].cpu().long()
        inv = can.invert_canonicalization(f.to(dev), induced_rep_type="scalar")
    chk = bench.oracle_check(can, x, f, y, inv, acts, gidx, group_type=group_type, num_rotations=N)
    # tolerances of test_headline_geometry_end_to_end_against_oracle_fft_path, the activation bound RELATIVE to their scale
    # (the random-init bench network gives activations of ~5e-3 on white noise)
    assert chk["acts_max_err"] <= 2e-5 * chk["acts_scale"], chk
    assert chk["index_match"] == 1.0, chk
    assert chk["canonicalize_max_err"] <= PIX_MAX and chk["canonicalize_rms_err"] <= PIX_RMS, chk
    assert chk["invert_max_err"] <= PIX_MAX and chk["invert_rms_err"] <= PIX_RMS, chk
    # the random-init bench network separates the orientations of white noise only weakly: the check above must not be vacuous
    assert chk["n_clear_margin"] >= 4, chk


def test_headline_whole_batch_of_256_against_oracle(dev):
    """BASELINE configs[1] at its own batch: the 256 images bench.py times (same seeds, same canonicalizer), every one of them
    through the CPU oracle -- activations, group index (100 % above the tie margin; the tie rate itself is bounded so the check
    cannot go vacuous), canonicalized and inverted pixels max + RMS.  This is bench.py's ``self_check`` as a test."""
    import bench

    can = bench.build_canonicalizer(dev)
    B = 256
    x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    f = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(1000))
    with torch.no_grad():
        y = can(x.to(dev))
        acts = can.canonicalization_info_dict["group_activations"].cpu()
        gidx = can.canonicalization_info_dict["group_index"].cpu().long()
        inv = can.invert_canonicalization(f.to(dev), induced_rep_type="scalar")
    chk = bench.oracle_check(can, x, f, y, inv, acts, gidx)
    assert chk["images"] == 256
    assert chk["acts_max_err"] <= 2e-5 * chk["acts_scale"], chk
    assert chk["index_match"] == 1.0, chk
    assert chk["tie_rate"] <= 0.25 and chk["n_clear_margin"] >= 192, chk
    assert chk["canonicalize_max_err"] <= PIX_MAX and chk["canonicalize_rms_err"] <= PIX_RMS, chk
    assert chk["invert_max_err"] <= PIX_MAX and chk["invert_rms_err"] <= PIX_RMS, chk
    # (the randomly initialised network prefers the right-angle elements on white noise -- its 45-degree banks are bilinear
    # re-samplings of the filters and respond more weakly; the 45-degree pixels are covered by test_canonicalize_headline_shape_error_budget)
    assert len(set(gidx.tolist())) >= 3, sorted(set(gidx.tolist()))


def _c4_regular_dense_layers(fields: int, k: int, layers: int, gen: torch.Generator):
    """A C4 regular-representation G-CNN in the EXPORTED dense form (what e2cnn's ``R2Conv.export()`` / ``InnerBatchNorm.export()``
    hand over: plain Conv2d over fields x |G| channels, channel index = field * 4 + element; plain BatchNorm2d with every
    per-field quantity repeated 4 times), built here from the group's regular representation with exact quarter turns --
    independently of the product's filter-bank code:
        lifting   W[(o, r), c]      = rot90^r( w[o, c] )
        regular   W[(o, r), (i, s)] = rot90^r( w[o, i, (s - r) mod 4] )
    so that rotating the input by a quarter turn rotates every output plane and shifts the element axis by one."""
    import torch.nn as nn

    convs, norms = [], []
    cin = 3
    for layer in range(layers):
        if layer == 0:
            w = torch.randn(fields, 3, k, k, generator=gen) * (1.0 / (3 * k * k)) ** 0.5
            W = torch.stack([torch.rot90(w, r, (-2, -1)) for r in range(4)], dim=1).reshape(fields * 4, 3, k, k)
        else:
            w = torch.randn(fields, fields, 4, k, k, generator=gen) * (1.0 / (fields * 4 * k * k)) ** 0.5
            W = torch.empty(fields, 4, fields, 4, k, k)
            for r in range(4):
                for s_ in range(4):
                    W[:, r, :, s_] = torch.rot90(w[:, :, (s_ - r) % 4], r, (-2, -1))
            W = W.reshape(fields * 4, fields * 4, k, k)
        cv = nn.Conv2d(cin, fields * 4, k, bias=True)
        with torch.no_grad():
            cv.weight.copy_(W)
            cv.bias.copy_((torch.randn(fields, generator=gen) * 0.1).repeat_interleave(4))
        convs.append(cv)
        cin = fields * 4
        if layer < layers - 1:
            bn = nn.BatchNorm2d(fields * 4)
            with torch.no_grad():
                bn.weight.copy_((torch.rand(fields, generator=gen) + 0.5).repeat_interleave(4))
                bn.bias.copy_((torch.randn(fields, generator=gen) * 0.2).repeat_interleave(4))
                bn.running_mean.copy_((torch.randn(fields, generator=gen) * 0.2).repeat_interleave(4))
                bn.running_var.copy_((torch.rand(fields, generator=gen) + 0.5).repeat_interleave(4))
            norms.append(bn.eval())
    return convs, norms


@pytest.mark.parametrize("fields,size", [(16, 96), (8, 60)])
def test_load_exported_dense_with_an_independently_built_regular_bank(dev, fields, size):
    """The bridge for e2cnn-trained weights (reference network: escnn_networks.py:48-91) fed with weights it did NOT export
    itself: a dense C4 regular-representation network built in this test from exact quarter turns + cyclic channel shifts.
    (1) the dense network is what it claims to be -- rotating the input by 90 degrees shifts its pooled activations by one
    element (plain torch, fp64); (2) the product loaded with those layers reproduces the plain-torch fp64 evaluation through
    its inference fast path (FFT / Winograd convolution, MFMA lifting layer, linearised last layer) and through its module
    path; (3) a canonicalizer built on it picks the element the dense network's own argmax picks."""
    import copy

    import equiadapt_amd as ea

    gen = torch.Generator().manual_seed(1234 + fields)
    k, layers, G = 5, 3, 4
    convs, norms = _c4_regular_dense_layers(fields, k, layers, gen)

    def dense_forward(x64):
        h = x64
        for i, cv in enumerate(convs):
            h = torch.nn.functional.conv2d(h, cv.weight.double(), cv.bias.double())
            if i < len(norms):
                bn = norms[i]
                h = (h - bn.running_mean.double()[None, :, None, None]) / torch.sqrt(bn.running_var.double()[None, :, None, None] + bn.eps)
                h = torch.relu(h * bn.weight.double()[None, :, None, None] + bn.bias.double()[None, :, None, None])
        return h.reshape(h.shape[0], fields, G, h.shape[-2], h.shape[-1]).mean(dim=(1, 3, 4))

    x = torch.randn(9, 3, size, size, generator=gen)
    with torch.no_grad():
        want = dense_forward(x.double())
        turned = dense_forward(torch.rot90(x, 1, (-2, -1)).double())
    assert (turned - torch.roll(want, 1, dims=1)).abs().max().item() <= 1e-12 * want.abs().max().item() + 1e-13, \
        "the test's own dense bank is not C4-equivariant"
    net = ea.ESCNNEquivariantNetwork((3, size, size), fields, k, "rotation", 4, layers).to(dev).eval()
    net.load_exported_dense([copy.deepcopy(c).to(dev) for c in convs], [copy.deepcopy(n).to(dev).eval() for n in norms])
    with torch.no_grad():
        fast = net(x.to(dev)).cpu().double()
    with torch.enable_grad():
        mod = net(x.to(dev)).detach().cpu().double()
    scale = want.abs().max().item()
    assert (fast - want).abs().max().item() <= 2e-5 * scale, ((fast - want).abs().max().item(), scale)
    assert (mod - want).abs().max().item() <= 2e-5 * scale
    top2 = want.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4 * scale
    assert int(clear.sum()) >= 5
    assert torch.equal(fast.argmax(-1)[clear], want.argmax(-1)[clear])
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=1.0, resize_shape=size)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, size, size)).to(dev).eval()
    with torch.no_grad():
        can(x.to(dev))
    assert torch.equal(can.canonicalization_info_dict["group_index"].cpu().long()[clear], want.argmax(-1)[clear])


@pytest.mark.parametrize("group_type,N", [("rotation", 8), ("roto-reflection", 4)])
def test_load_exported_dense_round_trip_and_fast_path(dev, group_type, N):
    """The bridge for e2cnn-trained weights (escnn_networks.py:48-91 export to Conv2d / BatchNorm2d): this network's own layers
    exported to the dense form and loaded into a second instance give the same activations -- through the inference fast
    path (FFT convolution on the 64-channel hidden layer, MFMA lifting layer, linearised tail), through the plain dense
    modules, and against the CPU oracle driven by the filter-bank state dict."""
    import equiadapt_amd as ea
    from equiadapt_amd.images.canonicalization_networks import fftconv
    from oracle import nets as onets

    torch.manual_seed(77)
    net = ea.ESCNNEquivariantNetwork((3, 96, 96), 8, 5, group_type, N, 3)
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.normal_(0.1, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    with torch.no_grad():
        for m in net.eqv_network:
            if hasattr(m, "expanded_weights"):
                m.bias.normal_(0, 0.1)
    net = net.to(dev).eval()
    G = net.num_group_elements
    convs, norms = net.export_dense()
    assert [c.weight.shape for c in convs] == [(8 * G, 3, 5, 5), (8 * G, 8 * G, 5, 5), (8 * G, 8 * G, 5, 5)]
    other = ea.ESCNNEquivariantNetwork((3, 96, 96), 8, 5, group_type, N, 3).to(dev).eval()
    other.load_exported_dense(convs, norms)
    x = torch.randn(9, 3, 96, 96)
    calls = []
    orig = fftconv.conv5x5
    fftconv.conv5x5 = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            want = net(x.to(dev)).cpu()
            n0 = len(calls)
            got = other(x.to(dev)).cpu()
        assert n0 == 1 and len(calls) == 2, "the dense form did not take the FFT fast path"
    finally:
        fftconv.conv5x5 = orig
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 1e-6 * scale, ((got - want).abs().max().item(), scale)
    with torch.enable_grad():                                             # plain dense modules (autograd on)
        mod = other(x.to(dev)).detach().cpu()
    assert (mod - want).abs().max().item() <= 2e-5 * scale
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    ref = onets.escnn_like_network(x, sd, group_type, N, 3, 8)
    assert (got - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1.0)
    # a per-channel bias that is NOT constant within a field (not what an equivariant export produces, but legal input)
    with torch.no_grad():
        convs[-1].bias.add_(torch.randn_like(convs[-1].bias))
    other.load_exported_dense(convs, norms)
    with torch.no_grad():
        fast = other(x.to(dev)).cpu()
    with torch.enable_grad():
        mod = other(x.to(dev)).detach().cpu()
    assert (fast - mod).abs().max().item() <= 2e-5 * max(mod.abs().max().item(), 1.0)
    with pytest.raises(ValueError):
        other.load_exported_dense(convs[:2], norms)
    with pytest.raises(ValueError):
        other.load_exported_dense([convs[0], convs[0], convs[2]], norms)


@pytest.mark.parametrize("M,Cin,Cout", [(36, 64, 64), (130, 32, 128), (64, 256, 256), (257, 96, 192)])
def test_cgemm3m_matches_real_gemm_in_fp64(dev, M, Cin, Cout):
    """The hand-written 3-multiplication complex GEMM on the fp32 MFMA (eqa_fft48k5_cgemm3m) against an fp64 evaluation of the
    real [M x 2Cin].[2Cin x 2Cout] product the GEMM library used to run, per stored frequency: ragged row counts (rows beyond M
    are clamped on load and never stored), one to four column tiles, 2 to 16 K-stages; the padding row of the pitched buffers
    stays untouched.  fp32 bound: the library's own fp32 result is 2-3e-7 of max|Mo| away from fp64; Karatsuba's imaginary part
    (a difference of three products) is allowed 8x that."""
    from equiadapt_amd import _lib
    from equiadapt_amd.images.canonicalization_networks import fftconv

    lib = _lib.load()
    assert lib.eqa_fft48k5_cgemm3m_supported(Cin, Cout) and not lib.eqa_fft48k5_cgemm3m_supported(48, 64) \
        and not lib.eqa_fft48k5_cgemm3m_supported(64, 32)
    g = torch.Generator().manual_seed(M + Cin)
    bank = (torch.randn(Cout, Cin, 5, 5, generator=g) / (5.0 * Cin ** 0.5)).to(dev)
    B = fftconv.filter_spectra(bank)                                    # (F, 2Cin, 2Cout) real form
    B3 = fftconv.filter_spectra3m(bank)
    V = fftconv.spectra_buffer(M, 2 * Cin, dev)
    V.copy_(torch.randn(fftconv.F, M, 2 * Cin, generator=g).to(dev))
    pitch = lib.eqa_fft48k5_tile_pitch(M)
    Mo = fftconv.contract(V, B3, M)
    assert Mo.shape == (fftconv.F, M, 2 * Cout) and Mo.stride(0) == pitch * 2 * Cout
    want = torch.bmm(V.double(), B.double())
    scale = want.abs().max().item()
    err = (Mo.double() - want).abs().max().item()
    lib_err = (torch.bmm(V, B).double() - want).abs().max().item()
    assert err <= 2.5e-6 * scale and err <= 8 * lib_err, (err, lib_err, scale)
    # imaginary and real parts separately (interleaved complex columns)
    assert (Mo.double() - want)[..., 0::2].abs().max().item() <= 6e-7 * scale
    # poison test: rows >= M of each frequency (the pitch padding) are not written
    full = torch.full((fftconv.F, pitch, 2 * Cout), 7.0, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.eqa_fft48k5_cgemm3m(V.data_ptr(), B3.data.data_ptr(), full.data_ptr(), M, Cin, Cout, st), "cgemm3m")
    assert torch.equal(full[:, :M], Mo) and (full[:, M:] == 7.0).all()
    # the convolution form (correlate = False) of the spectra as well: conj of the correlation form
    B2 = fftconv.filter_spectra(bank, correlate=False)
    Mo2 = fftconv.contract(V, fftconv.filter_spectra3m(bank, correlate=False), M)
    want2 = torch.bmm(V.double(), B2.double())
    assert (Mo2.double() - want2).abs().max().item() <= 2.5e-6 * want2.abs().max().item()


def test_fft48_convolution_3m_gemm_path_matches_conv2d(dev):
    """The FFT convolution with the hand-written 3-multiplication complex GEMM as its contraction (eqa_fft48k5_input ->
    eqa_fft48k5_cgemm3m -> eqa_fft48k5_output / _output_sums) against F.conv2d in fp64: exact tiles, ragged last tiles in both
    axes, a single tile, more than 64 tiles (two row tiles of the GEMM, the second ragged), bias / ReLU on both sides, both
    window-sum sizes."""
    import torch.nn.functional as F

    from equiadapt_amd.images.canonicalization_networks import fftconv

    torch.manual_seed(49)
    assert fftconv.GEMM == "3m"
    cases = [(2, 32, 64, 92, 92), (1, 64, 64, 60, 97), (1, 32, 64, 48, 48), (3, 32, 128, 137, 49), (1, 64, 64, 53, 50), (1, 32, 64, 50, 52),
             (20, 32, 64, 92, 92)]
    for (B, Cin, Cout, H, W) in cases:
        x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(Cout, Cin, 5, 5, device=dev) / (5 * Cin ** 0.5)
        b, ib = torch.randn(Cout, device=dev), torch.randn(Cin, device=dev)
        Bm = fftconv.spectra_for(w)
        assert isinstance(Bm, fftconv.Spectra3M)
        for (relu, in_relu) in [(False, False), (True, True)]:
            xin = torch.relu(x.double() + ib.double()[None, :, None, None]) if in_relu else x.double()
            want = F.conv2d(xin, w.double(), b.double())
            want = torch.relu(want) if relu else want
            got = fftconv.conv5x5(x, Bm, b, relu, ib if in_relu else None, in_relu)
            assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
            scale = want.abs().max().item()
            assert (got.double() - want).abs().max().item() <= 5e-6 * scale, (B, Cin, Cout, H, W, relu, (got.double() - want).abs().max().item() / scale)
            OH, OW = H - 4, W - 4
            for k in (5, 3):
                if min(OH, OW) < 2 * k - 1:
                    continue
                S = fftconv.conv5x5(x, Bm, b, relu, ib if in_relu else None, in_relu, sums_k=k)
                Sw = torch.stack([torch.stack([want[:, :, u:u + OH - k + 1, v:v + OW - k + 1].sum((-1, -2)) for v in range(k)], -1)
                                  for u in range(k)], -2)
                assert S.dtype == torch.float64 and ((S - Sw).abs().max() <= 2e-6 * Sw.abs().max().clamp_min(scale)), (B, H, W, k)


def test_conv_s2_mfma_matches_conv2d(dev):
    """eqa_conv_s2 (ConvNetwork's stride-2 convolutions as an implicit GEMM on the fp32 MFMA, bias + exact GELU in the epilogue,
    channels-last output) against F.conv2d / F.gelu in fp64: the planar first-layer form (1-4 input planes) and the channels-last
    form (16 / 32 / 64 channels), k = 3, 5, 7, padding 0 and 1, pixel counts that are not a multiple of the 256-pixel block."""
    import torch.nn.functional as F

    from equiadapt_amd import ops

    torch.manual_seed(50)
    cases = [(True, 3, 16, 7, 0, 5, 128, 128), (True, 1, 16, 3, 1, 3, 30, 41), (True, 3, 32, 5, 0, 2, 64, 64), (True, 4, 16, 5, 1, 1, 21, 19),
             (False, 16, 16, 7, 0, 5, 61, 61), (False, 16, 32, 7, 1, 3, 28, 28), (False, 32, 32, 5, 0, 2, 33, 29), (False, 32, 64, 3, 1, 2, 17, 23),
             (False, 64, 64, 5, 1, 1, 20, 20), (False, 16, 16, 3, 0, 1, 9, 9)]
    for (planar, Cin, Cout, K, pad, B, H, W) in cases:
        assert ops.conv_s2_supported(Cin, Cout, K, pad, planar), (planar, Cin, Cout, K, pad)
        x = torch.randn(B, Cin, H, W, device=dev)
        w = torch.randn(Cout, Cin, K, K, device=dev) / (K * Cin ** 0.5)
        b = torch.randn(Cout, device=dev)
        wp = ops.pack_conv_s2_weights(w, planar)
        xin = x.contiguous() if planar else x.permute(0, 2, 3, 1).contiguous()
        for gelu in (True, False):
            got = ops.conv_s2(xin, wp, b, gelu, Cout, K, pad, planar)
            want = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=pad)
            want = F.gelu(want) if gelu else want
            assert got.shape == (B, want.shape[2], want.shape[3], Cout)
            err = (got.permute(0, 3, 1, 2).double() - want).abs().max().item()
            assert err <= 2e-6 * max(want.abs().max().item(), 1.0), (planar, Cin, Cout, K, pad, gelu, err)
    assert not ops.conv_s2_supported(3, 8, 5, 0, True) and not ops.conv_s2_supported(16, 16, 4, 0, False) \
        and not ops.conv_s2_supported(16, 16, 5, 2, False) and not ops.conv_s2_supported(8, 16, 5, 0, False)


def test_graphed_canonicalizer_matches_eager(dev):
    """equiadapt_amd.graphs.GraphedCanonicalizer: the captured hipGraph of canonicalize + invert replays bit-identically to the
    eager step on new data, for an image canonicalizer (with invert) and the point-cloud one (no invert defined)."""
    import equiadapt_amd as ea
    from equiadapt_amd.graphs import GraphedCanonicalizer

    torch.manual_seed(23)
    net = ea.CustomEquivariantNetwork((3, 32, 32), 8, 5, "rotation", 4, 2, device="cpu")
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=1.0, resize_shape=32)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 32, 32)).to(dev)
    with pytest.raises(RuntimeError, match="eval"):
        GraphedCanonicalizer(can.train(), (16, 3, 32, 32))
    can.eval()
    step = GraphedCanonicalizer(can, (16, 3, 32, 32), (16, 3, 32, 32))
    for seed in (1, 2):
        x = torch.randn(16, 3, 32, 32, generator=torch.Generator().manual_seed(seed)).to(dev)
        f = torch.randn(16, 3, 32, 32, generator=torch.Generator().manual_seed(50 + seed)).to(dev)
        y, idx, inv = step(x, f)
        torch.cuda.synchronize()
        y, idx, inv = y.clone(), idx.clone(), inv.clone()
        with torch.no_grad():
            y_e = can(x)
            idx_e = can.canonicalization_info_dict["group_index"]
            inv_e = can.invert_canonicalization(f, induced_rep_type="scalar")
        assert torch.equal(y, y_e) and torch.equal(idx, idx_e) and torch.equal(inv, inv_e)
    with pytest.raises(ValueError, match="captured for"):
        step(torch.zeros(8, 3, 32, 32, device=dev))

    hp4 = types.SimpleNamespace(n_knn=20, pooling="mean")
    can4 = ea.EquivariantPointcloudCanonicalization(ea.VNSmall(hp4), hp4).to(dev).eval()
    step4 = GraphedCanonicalizer(can4, (4, 3, 128))
    pc = torch.randn(4, 3, 128, device=dev)
    y, R, inv = step4(pc)
    torch.cuda.synchronize()
    with torch.no_grad():
        y_e = can4(pc)
    assert inv is None and torch.equal(y, y_e) and torch.equal(R, can4.canonicalization_info_dict["group_element"]["rotation"])


def test_graphed_canonicalizer_with_targets_matches_eager(dev):
    """The COCO-shaped step (optimised D4 canonicalizer + ConvNetwork, uint8 masks and boxes as targets, scalar invert) captured as
    one hipGraph: image, masks, boxes and the inverted output replay bit-identically to the eager step on new data
    (reference: discrete_group.py:190-259 with the targets branch :217-236)."""
    import equiadapt_amd as ea
    from equiadapt_amd.graphs import GraphedCanonicalizer

    torch.manual_seed(29)
    net = ea.ConvNetwork((3, 32, 32), out_channels=16, kernel_size=3, num_layers=2, out_vector_size=16)
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=1.0, resize_shape=32, group_type="roto-reflection", num_rotations=4,
                               artifact_err_wt=0.0, learn_ref_vec=False)
    can = ea.OptimizedGroupEquivariantImageCanonicalization(net, hp, (3, 64, 64)).to(dev).eval()
    B = 5

    def make(seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, 3, 64, 64, generator=g).to(dev)
        f = torch.randn(B, 1, 64, 64, generator=g).to(dev)
        t = [{"boxes": (torch.rand(2, 4, generator=g) * 30 + torch.tensor([0.0, 0.0, 30.0, 30.0])).to(dev),
              "masks": (torch.rand(2, 64, 64, generator=g) > 0.5).to(torch.uint8).to(dev)} for _ in range(B)]
        return x, f, t

    x, f, t = make(0)
    step = GraphedCanonicalizer(can, x.shape, f.shape, targets_like=t)
    for seed in (1, 2):
        x, f, t = make(seed)
        boxes_before = [d["boxes"].clone() for d in t]
        y, idx, inv = step(x, f, t)
        torch.cuda.synchronize()
        got = (y.clone(), idx.clone(), inv.clone(), [{k: v.clone() for k, v in d.items()} for d in step.targets])
        assert all(torch.equal(a, d["boxes"]) for a, d in zip(boxes_before, t)), "the caller's boxes are copied, not flipped in place"
        with torch.no_grad():
            y_e, t_e = can(x, [{k: v.clone() for k, v in d.items()} for d in t])
            inv_e = can.invert_canonicalization(f, induced_rep_type="scalar")
        assert torch.equal(got[0], y_e) and torch.equal(got[1], can.canonicalization_info_dict["group_index"]) and torch.equal(got[2], inv_e)
        for a, b in zip(got[3], t_e):
            assert torch.equal(a["masks"], b["masks"]) and torch.equal(a["boxes"], b["boxes"])
    with pytest.raises(ValueError, match="target"):
        step(x, f, [{"boxes": d["boxes"][:1], "masks": d["masks"]} for d in t])


def test_lift_conv_dense_forms_agree_on_random_shapes(dev):
    """The dense form of the lifting convolution (5 x 5 over RGB, 64-channel slices, rows >= 32 pixels: K = 76, tiles over the flattened
    map) in its three outputs -- channels-last, channel-group-major, training form with running sums -- on random shapes whose tiles
    cross row ends at every offset: the three hold the same bits, the values match F.conv2d in fp64, the sums match the map."""
    import random

    import torch.nn.functional as F

    from equiadapt_amd import ops
    from equiadapt_amd.images.canonicalization_networks import fftconv

    rng = random.Random(5)
    torch.manual_seed(5)
    for _ in range(12):
        B, H, W = rng.randint(1, 9), rng.randint(5, 60), rng.randint(36, 130)
        Cout = rng.choice([64, 128, 256])
        x = (torch.randn(B, 3, H, W, device=dev) + 0.2).contiguous(memory_format=torch.channels_last)
        bank = torch.randn(Cout, 3, 5, 5, device=dev) / 8
        bias = torch.randn(Cout, device=dev)
        wpk = ops.pack_lift_weights(bank)
        y = ops.lift_conv_nhwc(x, wpk, bias, True, 5, 5)
        want = torch.relu(F.conv2d(x.double(), bank.double(), bias.double()))
        assert (y.double() - want).abs().max().item() <= 2e-6 * want.abs().max().item(), (B, H, W, Cout)
        g = fftconv.GroupedMap(ops.lift_conv_grouped(x, wpk, bias, True, 5, 5))
        assert torch.equal(g.to_channels_last(), y), (B, H, W, Cout)
        assert ops.lift_conv_stats_supported(x.shape, 5, 5, Cout)
        y2, part = ops.lift_conv_nhwc_stats(x, wpk, 5, 5)
        assert torch.equal(y2, ops.lift_conv_nhwc(x, wpk, None, False, 5, 5)), (B, H, W, Cout)
        exact = y2.permute(0, 2, 3, 1).reshape(-1, Cout).double()
        truth = torch.stack([exact.sum(0), (exact * exact).sum(0)], dim=1)
        assert (part.sum(0) - truth).abs().max().item() <= 1e-5 * truth[:, 1].max().item(), (B, H, W, Cout)


def test_grouped_activation_layout_between_lift_and_fft(dev):
    """The lifting convolution's channel-group-major output (eqa_lift_conv_grouped) holds exactly the channels-last result, and the
    FFT convolution reads it (eqa_fft48k5_input_grouped) to exactly the same spectra / output -- the layout changes which bytes sit
    next to each other, not one arithmetic operation.  Also a width whose last tile is partial (general load path)."""
    from equiadapt_amd import ops
    from equiadapt_amd.images.canonicalization_networks import fftconv

    torch.manual_seed(41)
    for (B, H, W) in ((8, 52, 96), (8, 64, 70)):
        x = torch.randn(B, 3, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        bank = torch.randn(64, 3, 5, 5, device=dev) / 8
        bias = torch.randn(64, device=dev)
        wpk = ops.pack_lift_weights(bank)
        y = ops.lift_conv_nhwc(x, wpk, bias, True, 5, 5)
        g = fftconv.GroupedMap(ops.lift_conv_grouped(x, wpk, bias, True, 5, 5))
        assert g.shape == y.shape
        assert torch.equal(g.to_channels_last(), y)
        w2 = torch.randn(64, 64, 5, 5, device=dev) / 40
        b2 = torch.randn(64, device=dev)
        Bf = fftconv.spectra_for(w2)
        for in_bias in (None, b2):
            o1 = fftconv.conv5x5(y, Bf, b2, True, in_bias, in_bias is not None)
            o2 = fftconv.conv5x5(g, Bf, b2, True, in_bias, in_bias is not None)
            assert torch.equal(o1, o2)
        s1 = fftconv.conv5x5(y, Bf, b2, True, sums_k=5)
        s2 = fftconv.conv5x5(g, Bf, b2, True, sums_k=5)
        assert torch.equal(s1, s2)


@pytest.mark.parametrize("induced_rep,num_channels", [("regular", 12), ("scalar", 3)])
def test_reference_own_test_invert_canonicalization_induced_rep(dev, induced_rep, num_channels):
    """The reference's own test of this path (tests/images/canonicalization/test_discrete_group.py:44-86) through the product: the
    same constructor arguments (ESCNNEquivariantNetwork((3,64,64), 32, k=3, C4, 2 layers), crop 0.9, resize (32,32), beta 0.1,
    hyper-parameters as a mapping like its DictConfig), one (1,3,64,64) image, then invert_canonicalization of a 12-channel
    regular / 3-channel scalar map: the shape the reference asserts, and the values against the oracle for the element chosen."""
    import equiadapt_amd as ea

    torch.manual_seed(0)
    net = ea.ESCNNEquivariantNetwork(in_shape=(3, 64, 64), out_channels=32, kernel_size=3, group_type="rotation", num_rotations=4,
                                     num_layers=2)
    hp = {"input_crop_ratio": 0.9, "resize_shape": (32, 32), "beta": 0.1}
    dgic = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 64, 64)).to(dev)
    image = torch.randn((1, 3, 64, 64))
    _ = dgic(image.to(dev))                                      # populates canonicalization_info_dict
    canonicalized_image = torch.randn((1, num_channels, 64, 64))
    inverted = dgic.invert_canonicalization(canonicalized_image.to(dev), **{"induced_rep_type": induced_rep})
    assert inverted.shape == canonicalized_image.shape           # what the reference's test asserts
    el = dgic.canonicalization_info_dict["group_element"]
    want = io.invert_action(canonicalized_image, el["rotation"].detach().cpu(), None, 4, 4, induced_rep)
    _close(inverted.detach().cpu(), want)


@pytest.mark.parametrize("group_type,N,out_ch,layers,res,B", [
    ("rotation", 4, 32, 3, 96, 8),          # C4: 128 channels, FFT layer behind a channel-group-major lifting layer
    ("rotation", 8, 16, 4, 96, 8),          # four layers: two FFT layers in a row (plain-output inverse feeding the next transform)
    ("roto-reflection", 4, 16, 3, 100, 8),  # D4, ragged tiles (96 -> 92 outputs: partial last tile)
    ("rotation", 8, 32, 3, 64, 12),         # 64-pixel input: 60 -> 56 outputs, tiles do not fit -> Winograd path
    ("rotation", 4, 8, 3, 96, 4),           # narrow network (32 channels) and too few tiles for the FFT path
])
def test_escnn_network_inference_paths_match_oracle_sweep(dev, group_type, N, out_ch, layers, res, B):
    """ESCNNEquivariantNetwork.forward in eval mode -- whichever fast path its shape selects (FFT / Winograd convolutions, MFMA
    lifting layer in either output layout, linearised tail) -- against the oracle's op-by-op restatement with the same state
    dict: group activations to 2e-5 of their scale (escnn_networks.py:93-117)."""
    import equiadapt_amd as ea
    from oracle import nets as onets

    torch.manual_seed(res + layers + out_ch)
    net = ea.ESCNNEquivariantNetwork((3, res, res), out_channels=out_ch, kernel_size=5, group_type=group_type, num_rotations=N,
                                     num_layers=layers)
    with torch.no_grad():
        for m in net.modules():                      # non-trivial batch-norm statistics (eval mode folds them)
            if hasattr(m, "running_mean") and m.running_mean is not None:
                m.running_mean.uniform_(-0.05, 0.05)
                m.running_var.uniform_(0.8, 1.2)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to(dev).eval()
    x = torch.randn(B, 3, res, res)
    with torch.no_grad():
        got = net(x.to(dev)).cpu()
        want = onets.escnn_like_network(x, sd, group_type, N, layers, out_ch)
    assert got.shape == want.shape
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 2e-5 * scale, (got - want).abs().max().item() / scale


@pytest.mark.parametrize("k,nimg,cin,cout,hw", [(9, 6, 64, 64, (56, 56)), (9, 3, 32, 64, (60, 97)), (7, 4, 64, 128, (54, 48)), (3, 5, 64, 64, (50, 93)),
                                                 (3, 2, 24, 40, (20, 31)), (7, 2, 16, 16, (90, 17)), (9, 2, 8, 12, (48, 49))])
def test_fft_convolution_any_kernel_size_matches_conv2d_with_gradients(dev, k, nimg, cin, cout, hw):
    """eqa_fft48_* (kernel sizes 3 / 7 / 9: 46 / 42 / 40 outputs per 48 x 48 tile): forward, input gradient and filter gradient of
    y = conv2d(x, bank) against an fp64 convolution + autograd -- channel counts with (64 / 64, 64 / 128, 32 / 64) and without
    (24 / 40, 16 / 16, 8 / 12) the hand-written complex GEMM, maps that one tile covers, that need partial last tiles, and whose
    tile grid is wider than high; previous layer's bias + ReLU on the loads and this layer's on the way out (inference form).
    Reference arithmetic: the dense conv2d of R2Conv for any kernel_size (escnn_networks.py:19-91)."""
    from equiadapt_amd.images.canonicalization_networks import fftconv

    torch.manual_seed(k * 100 + cin)
    H, W = hw
    x = torch.randn(nimg, cin, H, W)
    bank = torch.randn(cout, cin, k, k) * (1.0 / (cin * k * k)) ** 0.5
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = bank.to(dev).requires_grad_(True)
    y = fftconv.ConvKxKFunction.apply(xd, bd)
    x64, b64 = x.double().requires_grad_(True), bank.double().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, b64)
    assert y.shape == y64.shape
    sy = y64.abs().max().item()
    assert (y.detach().cpu().double() - y64.detach()).abs().max().item() <= 2e-6 * sy
    g = torch.randn(y64.shape)
    y.backward(g.to(dev))
    y64.backward(g.double())
    assert (xd.grad.cpu().double() - x64.grad).abs().max().item() <= 4e-6 * x64.grad.abs().max().item()
    assert (bd.grad.cpu().double() - b64.grad).abs().max().item() <= 1e-5 * b64.grad.abs().max().item()
    # inference form: act(x) = relu(x + in_bias) on the loads, bias + ReLU on the output
    ib, ob = torch.randn(cin) * 0.3, torch.randn(cout) * 0.3
    with torch.no_grad():
        got = fftconv.conv_kxk(xd.detach(), fftconv.spectra_for_k(bd.detach()), k, ob.to(dev), True, ib.to(dev), True).cpu().double()
        want = torch.relu(torch.nn.functional.conv2d(torch.relu(x.double() + ib.double()[None, :, None, None]), bank.double())
                          + ob.double()[None, :, None, None])
    assert (got - want).abs().max().item() <= 3e-6 * want.abs().max().item()


@pytest.mark.parametrize("group_type,N,out_ch,k,layers,res,B", [
    ("rotation", 4, 16, 9, 3, 64, 16),         # the reference tutorial's canonicalization network (cell 17): FFT hidden layer, k = 9 window-sum tail
    ("rotation", 4, 16, 7, 3, 64, 8),
    ("rotation", 8, 32, 3, 3, 96, 8),          # k = 3 at 256 channels: the cost model picks the FFT path
    ("rotation", 4, 8, 3, 3, 40, 4),           # the reference's own test configuration (tests/images/canonicalization/test_discrete_group.py:31-38): library path
    ("roto-reflection", 4, 8, 9, 2, 48, 8),    # two layers: lifting layer straight into the k = 9 tail
])
def test_escnn_network_other_kernel_sizes_match_oracle(dev, group_type, N, out_ch, k, layers, res, B):
    """ESCNNEquivariantNetwork with kernel sizes other than 5 (the constructor argument is free in the reference,
    escnn_networks.py:19-44): eval-mode activations against the oracle's op-by-op restatement, and the training step's
    activations + parameter gradients through the fast path against the module path (autograd through conv2d)."""
    import copy

    import equiadapt_amd as ea
    from oracle import nets as onets

    torch.manual_seed(res + k)
    net = ea.ESCNNEquivariantNetwork((3, res, res), out_channels=out_ch, kernel_size=k, group_type=group_type, num_rotations=N, num_layers=layers)
    with torch.no_grad():
        for m in net.modules():
            if hasattr(m, "running_mean") and m.running_mean is not None:
                m.running_mean.uniform_(-0.05, 0.05)
                m.running_var.uniform_(0.8, 1.2)
    sd = {kk: v.detach().clone() for kk, v in net.state_dict().items()}
    net = net.to(dev).eval()
    x = torch.randn(B, 3, res, res)
    with torch.no_grad():
        got = net(x.to(dev)).cpu()
        want = onets.escnn_like_network(x, sd, group_type, N, layers, out_ch)
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 2e-5 * scale, (got - want).abs().max().item() / scale
    # training: fast path vs module path, dropout off so that the two draw no different masks
    fast, slow = copy.deepcopy(net).train(), copy.deepcopy(net).train()
    for m in list(fast.modules()) + list(slow.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    wgt = torch.randn(B, net.num_group_elements, device=dev)
    a_fast = fast(x.to(dev))
    (a_fast * wgt).sum().backward()
    import os
    os.environ["EQA_TRAIN_FAST"] = "0"
    try:
        a_slow = slow(x.to(dev))
        (a_slow * wgt).sum().backward()
    finally:
        del os.environ["EQA_TRAIN_FAST"]
    assert (a_fast - a_slow).abs().max().item() <= 5e-5 * a_slow.abs().max().item()
    gscale = max(p.grad.abs().max().item() for p in slow.parameters() if p.grad is not None)
    for (n1, p1), (_, p2) in zip(fast.named_parameters(), slow.named_parameters()):
        if p2.grad is None:
            continue
        assert p1.grad is not None, n1
        # (a hidden layer's convolution bias cancels inside the batch-norm behind it: the module path leaves rounding noise there,
        # the fast path an exact zero -- hence the floor relative to the largest gradient of the network)
        assert (p1.grad - p2.grad).abs().max().item() <= 2e-3 * max(p2.grad.abs().max().item(), 1e-2 * gscale), \
            (n1, (p1.grad - p2.grad).abs().max().item(), p2.grad.abs().max().item())


@pytest.mark.parametrize("group_type,N,C,Cf,rep,hw", [("rotation", 8, 3, 3, "scalar", (224, 224)), ("rotation", 8, 3, 8, "regular", (64, 64)),
                                                      ("roto-reflection", 4, 2, 8, "regular", (40, 56)), ("roto-reflection", 4, 3, 6, "scalar", (33, 45)),
                                                      ("rotation", 4, 1, 5, "scalar", (30, 30))])
def test_group_action_pair_is_bit_identical_to_the_two_launches(dev, group_type, N, C, Cf, rep, hw):
    """eqa_group_action_pair: canonicalize x and invert f for the same group index in ONE launch (two jobs on one tile grid),
    or two library-issued launches when the jobs' channel staging widths differ -- either way bit-identical to
    eqa_canon_transform_fwd + eqa_invert_action_fwd (discrete_group.py:204-215 / images/utils.py:54-89 back to back), and
    within the pixel tolerance of the oracle."""
    from equiadapt_amd import ops
    from equiadapt_amd.images.utils import device_tables
    from oracle import image_ops as io

    H, W = hw
    refl = group_type == "roto-reflection"
    G = 2 * N if refl else N
    B = 11
    torch.manual_seed(31)
    x, f = torch.randn(B, C, H, W), torch.randn(B, Cf, H, W)
    gidx = torch.randint(0, G, (B,), dtype=torch.int32)
    pad = 0 if C == 1 else math.ceil(W / 2) if H == W else math.ceil(max(H, W) / 2)
    frame = (H + 2 * pad, W + 2 * pad)
    th_c, fl_c = device_tables("canonicalize", N, refl, frame, dev)
    th_i, fl_i, cmap = device_tables("invert", N, refl, (H, W), dev)
    cm = cmap if rep == "regular" else None
    xd, fd, gd = x.to(dev), f.to(dev), gidx.to(dev)
    y1 = ops.canon_transform(xd, gd, th_c, fl_c, pad)
    o1 = ops.invert_action(fd, gd, th_i, fl_i, cm)
    y2, o2 = ops.group_action_pair(xd, fd, gd, th_c, fl_c, pad, th_i, fl_i, cm)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2) and torch.equal(o1, o2)
    if H == W:   # the oracle's pad rule (Pad(ceil(W/2))) as the reference applies it to square images
        ang = io.group_angles(N)
        rot = (torch.cat([ang, ang]) if refl else ang)[gidx.long()]
        rf = (gidx >= N).float() if refl else None
        assert (y2.cpu() - io.canonicalize_images(x, rot, rf, (C, H, W))).abs().max().item() <= 1e-3
        assert (o2.cpu() - io.invert_action(f, rot, rf, N, G, rep)).abs().max().item() <= 1e-3


@pytest.mark.parametrize("pooling", ["mean", "max"])
def test_fused_vnsmall_any_k_and_both_kernels_match_the_op_path(dev, pooling):
    """eqa_vnsmall_fwd for neighbourhood sizes 1..32 (four lanes per point, distributed sorted list; k <= 20 and k <= 32
    instantiations), ragged cloud sizes (N not a multiple of the 64 points of a block or of the 16-candidate scan step, N == k), and
    the one-thread-per-point kernel (k = 20, eqa_set_option key 1) -- each against the op-by-op module path of the same network
    (equivariant_networks.py:15-76, 128-150)."""
    import equiadapt_amd as ea
    from equiadapt_amd import _lib

    lib = _lib.load()
    torch.manual_seed(40)
    tol = 3e-6 if pooling == "mean" else 2e-5
    for k, B, N in [(1, 2, 40), (3, 2, 70), (8, 3, 300), (16, 2, 513), (19, 2, 64), (20, 3, 1024), (20, 1, 20), (21, 2, 100), (27, 2, 1000),
                    (32, 2, 32), (32, 2, 777)]:
        net = ea.VNSmall(types.SimpleNamespace(n_knn=k, pooling=pooling))
        # k = 1: the only neighbour is the point itself, the edge feature is [0, x, 0], every channel's q and gate direction d are
        # parallel, and a closed gate leaves q - <q, d>/|d|^2 d = rounding noise of size 1e-8 -- which a batch-norm SHIFT then
        # divides by |q| + 1e-6: the network amplifies last-bit differences to 1e-2.  Only with the initial statistics (shift 0) is
        # that configuration a meaningful comparison.
        for m in (net.modules() if k > 1 else ()):
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.normal_(0.5, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.2)
        net = net.to(dev).eval()
        x = torch.randn(B, 3, N, device=dev)
        with torch.enable_grad():
            slow = net(x).detach()                     # grad mode -> op-by-op torch path
        scale = max(slow.abs().max().item(), 1.0)
        with torch.no_grad():
            quad = net(x)
        assert (quad - slow).abs().max().item() <= tol * scale, (k, B, N, (quad - slow).abs().max().item())
        if k == 20:
            assert lib.eqa_set_option(1, 1) == 0 and lib.eqa_get_option(1) == 1
            try:
                with torch.no_grad():
                    single = net(x)
            finally:
                assert lib.eqa_set_option(1, 0) == 0
            assert (single - slow).abs().max().item() <= tol * scale, (k, B, N)
    assert lib.eqa_set_option(1, 3) == -1


@pytest.mark.parametrize("group_type,N", [("rotation", 4), ("roto-reflection", 4), ("rotation", 8)])
def test_custom_network_lifting_conv_on_the_mfma_kernel(dev, group_type, N, monkeypatch):
    """CustomEquivariantNetwork's one convolution (the Z2 -> G lifting layer, custom_group_equivariant_layers.py:92-111 /
    :190-226 of the reference) on the hand-written fp32-MFMA kernel in inference: the CIFAR-shaped configuration (3 -> 8 fields,
    k = 5, two layers) against the oracle network, and the layer alone against the framework's convolution."""
    import torch.nn.functional as F

    import equiadapt_amd as ea
    from equiadapt_amd import ops
    from oracle import nets as onets

    torch.manual_seed(17)
    O = 8 if group_type == "rotation" else 4                 # O * |G| = 32 or 64 channels
    net = ea.CustomEquivariantNetwork((3, 32, 32), O, 5, group_type, N, 2, device="cpu").to(dev).eval()
    with torch.no_grad():
        net.eqv_network[0].bias.normal_(0, 0.1)
    x = torch.randn(9, 3, 32, 32, device=dev)
    lift = net.eqv_network[0]
    assert lift.mfma_lifting_ok(x) is False                   # grad mode: the module path
    with torch.no_grad():
        assert lift.mfma_lifting_ok(x)
        calls = []
        orig = ops.lift_conv_nhwc
        monkeypatch.setattr(ops, "lift_conv_nhwc", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        fast = net(x)
        y5 = lift(x)                                          # the layer alone, 5-D as the reference returns it
        assert len(calls) == 2, "the inference paths must run the hand-written kernel"
    want5 = F.conv2d(x.double(), lift.expanded_weights().double()).reshape(9, O, lift.num_group_elements, 28, 28) \
        + lift.bias.double()[None, :, None, None, None]
    assert (y5.double() - want5).abs().max().item() <= 2e-6 * want5.abs().max().item()
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    want = onets.custom_equivariant_network(x.cpu(), sd, group_type, N, 2)
    assert torch.allclose(fast.cpu(), want, atol=2e-5, rtol=1e-3)


def test_plane_gemm_matches_the_batched_product(dev):
    """eqa_plane_gemm (the Winograd planes' channel contraction on the fp32 MFMA, csrc/planegemm.hip) vs the fp64 batched product:
    36 and 64 planes, tile counts off the 64-row wave tile (1, 63, 65, 1000), 32 / 64 / 96 / 256 channels (one and two
    32-column subtiles per wave), rows beyond the tile count untouched.  fp32 fmaf chains over K <= 256: 1e-6 of the scale."""
    from equiadapt_amd import _lib, ops

    torch.manual_seed(60)
    for (T, P, Cin, Cout) in [(1, 36, 32, 32), (63, 64, 64, 96), (65, 36, 96, 64), (1000, 64, 128, 128), (300, 64, 256, 256), (130, 36, 32, 160)]:
        assert ops.plane_gemm_supported(Cin, Cout)
        V = torch.randn(T + 3, P, Cin, device=dev)
        U = torch.randn(P, Cin, Cout, device=dev) / Cin ** 0.5
        M = torch.full((T + 3, P, Cout), 7.0, device=dev)
        ops.plane_gemm(V, ops.pack_plane_gemm_weights(U), M, T)
        want = torch.einsum("tpk,pkn->tpn", V[:T].double(), U.double())
        assert (M[:T].double() - want).abs().max().item() <= 1e-6 * want.abs().max().item(), (T, P, Cin, Cout)
        assert (M[T:] == 7.0).all(), "rows beyond the tile count must not be written"
    assert not ops.plane_gemm_supported(48, 64) and not ops.plane_gemm_supported(64, 48)
    lib = _lib.load()
    assert lib.eqa_plane_gemm(V.data_ptr(), U.data_ptr(), M.data_ptr(), 10, 36, 48, 64, None) == -3


def test_fft_forward_pipeline_matches_the_one_block_per_item_kernel(dev, monkeypatch):
    """fft48_fwd_pipe_kernel (EQA_FFT_FWD_PIPE=1: persistent row / column waves, the two real edge columns packed into one complex
    transform) against the default fft48_fwd_fused_kernel on the same channel-group-major map: the convolution's output within 1e-6 of its
    scale (the packed edge columns round differently), and both within the FFT path's 2e-6 of an fp64 convolution."""
    import torch.nn.functional as F

    from equiadapt_amd import ops
    from equiadapt_amd.images.canonicalization_networks import fftconv

    torch.manual_seed(44)
    B, C = 64, 128
    x = torch.randn(B, 3, 96, 96, device=dev).contiguous(memory_format=torch.channels_last)
    bank = torch.randn(C, 3, 5, 5, device=dev) / 8
    g = fftconv.GroupedMap(ops.lift_conv_grouped(x, ops.pack_lift_weights(bank), None, True, 5, 5))
    assert fftconv.applicable(g, C, C)
    w2 = torch.randn(C, C, 5, 5, device=dev) / 60
    b2 = torch.randn(C, device=dev)
    Bf = fftconv.spectra_for(w2)
    o_item = fftconv.conv5x5(g, Bf, b2, True)
    monkeypatch.setenv("EQA_FFT_FWD_PIPE", "1")          # opt-in: measured slower than the default kernel (csrc/fftconv.hip)
    o_pipe = fftconv.conv5x5(g, Bf, b2, True)
    monkeypatch.delenv("EQA_FFT_FWD_PIPE")
    scale = o_item.abs().max().item()
    assert (o_pipe - o_item).abs().max().item() <= 1e-6 * scale
    assert not torch.equal(o_pipe, o_item), "the two kernels round the edge columns differently: identical output means the pipeline did not run"
    y = g.to_channels_last()[:4]
    want = torch.relu(F.conv2d(y.double(), w2.double(), b2.double()))
    assert (o_pipe[:4].double() - want).abs().max().item() <= 2e-6 * want.abs().max().item()


def test_invert_detection_outputs_matches_the_reference_loop(dev):
    """inference.invert_detection_outputs: the box inversion of the reference's segmentation GroupInference.forward
    (examples/images/segmentation/inference_utils.py:86-117: rotate_boxes(+degree) then flip_boxes where the element reflects; labels /
    scores / masks passed through) for a whole batch in one launch, against the oracle's per-sample loop."""
    from equiadapt_amd.inference import invert_detection_outputs

    torch.manual_seed(71)
    B, W = 9, 200
    rot = torch.tensor([0.0, 90.0, 180.0, 270.0, 45.0, 315.0, 90.0, 0.0, 180.0])
    refl = torch.tensor([0.0, 1.0, 0.0, 1.0, 1.0, 0.0, 0.0, 1.0, 1.0])
    outs = []
    for i in range(B):
        n = [3, 0, 1, 5, 2, 4, 1, 2, 3][i]
        xy = torch.rand(n, 2) * 100
        boxes = torch.cat([xy, xy + torch.rand(n, 2) * 90 + 1], dim=1)
        outs.append({"boxes": boxes, "labels": torch.arange(n), "scores": torch.rand(n), "masks": (torch.rand(n, 8, 8) > 0.5)})
    for with_reflection in (True, False):
        can = types.SimpleNamespace(canonicalization_info_dict={"group_element": {"rotation": rot.to(dev), **({"reflection": refl.to(dev)} if with_reflection else {})}})
        got = invert_detection_outputs(can, [{k: v.to(dev) for k, v in o.items()} for o in outs], W)
        for i, (g, o) in enumerate(zip(got, outs)):
            want = io.rotate_boxes(o["boxes"].clone(), rot[i], W)
            if with_reflection and refl[i]:
                want = io.flip_boxes(want, W)
            assert g["boxes"].shape == want.shape
            assert torch.allclose(g["boxes"].cpu(), want, atol=2e-4), (with_reflection, i)
            assert torch.equal(g["labels"].cpu(), o["labels"]) and torch.equal(g["masks"].cpu(), o["masks"]) and torch.equal(g["scores"].cpu(), o["scores"])
    ident = types.SimpleNamespace(canonicalization_info_dict={})
    assert invert_detection_outputs(ident, outs, W) is outs


@pytest.mark.parametrize("pooling,k,B,N", [("mean", 20, 64, 1024), ("max", 20, 5, 1000), ("mean", 8, 3, 333), ("mean", 32, 2, 4096)])
def test_fused_pointcloud_canonicalize_equals_the_three_kernels(dev, pooling, k, B, N):
    """eqa_vnsmall_canonicalize (network kernel + ONE tail kernel: output vectors, Gram-Schmidt frame, rotated cloud) against
    eqa_vnsmall_fwd -> eqa_gram_schmidt -> eqa_so3_rotate on the same clouds, and through the canonicalizer class (same info dict)."""
    import equiadapt_amd as ea
    from equiadapt_amd import ops

    torch.manual_seed(N + k)
    hp = types.SimpleNamespace(n_knn=k, pooling=pooling)
    net = ea.VNSmall(hp).to(dev).eval()
    x = torch.randn(B, 3, N, device=dev)
    prm = net.packed_parameters()
    with torch.no_grad():
        vec0 = ops.vnsmall_forward(x, prm, k, pooling)
        R0 = ops.gram_schmidt(vec0)
        y0 = ops.so3_rotate(x, R0)
        vec, R, y = ops.vnsmall_canonicalize(x, prm, k, pooling)
        assert torch.equal(vec, vec0)
        assert (R - R0).abs().max().item() <= 1e-6 and (y - y0).abs().max().item() <= 1e-5
        can = ea.EquivariantPointcloudCanonicalization(net, hp).to(dev).eval()
        yc = can(x)
        assert torch.equal(yc, y)
        assert torch.equal(can.canonicalization_info_dict["group_element_matrix_representation"], R)
        assert can.canonicalization_info_dict["group_element"]["rotation"] is can.canonicalization_info_dict["group_element_matrix_representation"]
    assert ops.vnsmall_canonicalize(torch.empty(0, 3, N, device=dev), prm, k, pooling)[2].shape == (0, 3, N)
