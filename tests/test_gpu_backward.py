"""GPU: gradients of the HIP transform / invert against torch autograd through the CPU oracle's op chain.

The oracle's kornia restatement is built from differentiable torch ops (rotation matrix -> inverse -> affine_grid ->
grid_sample), so `loss.backward()` on it yields exactly what the reference's autograd would: d/d rotation (the path by
which the task loss trains the canonicalizer), d/d reflection indicator, d/d input.
Tolerances: gradients are sums of O(1e5) terms of fp32 products -> relative 2e-3 of the gradient's own scale.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import image_ops as io  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _check_angle_grad(got, ref, angles_deg):
    """d/d angle parity.  Off the pixel grid: strict (2e-3 of the gradient scale).  At multiples of 90 deg every
    sample point sits ON the grid, where the bilinear interpolant has a kink: which one-sided derivative a pixel
    contributes is decided by the last ulp of its coordinate -- in the reference as well (and at a zero-padded
    border the two sides differ by the full pixel value).  There the reference's own gradient is one arbitrary
    subgradient, so only sign/magnitude sanity is asserted (35 % of the gradient scale); 0 and 180 deg agree to
    1e-6 in practice because sin is exactly / nearly zero there."""
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    kink = (angles_deg % 90.0) == 0
    scale = ref.abs().max() + 1e-12
    if (~kink).any():
        assert ((got - ref)[~kink].abs().max() / scale).item() < 2e-3
    if kink.any():
        assert ((got - ref)[kink].abs().max() / scale).item() < 0.35


def _smooth(B, C, H, W, seed):
    """Low-frequency images with per-(sample, channel) phases and amplitudes."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    ph = torch.rand(B, C, 1, 1, generator=g) * 6.28
    fx = 0.05 + 0.1 * torch.rand(B, C, 1, 1, generator=g)
    fy = 0.05 + 0.1 * torch.rand(B, C, 1, 1, generator=g)
    return (torch.sin(fx * xx + ph) * torch.cos(fy * yy - ph) + 0.02 * xx - 0.01 * yy).contiguous()


@pytest.mark.parametrize("group_type,N", [("rotation", 8), ("roto-reflection", 4)])
@pytest.mark.parametrize("shape", [(3, 40, 40), (2, 36, 52)])
def test_canonicalize_gradients(dev, group_type, N, shape):
    from equiadapt_amd.images.canonicalization.discrete_group import _CanonTransformFn
    from equiadapt_amd.images.utils import device_tables

    C, H, W = shape
    G = N if group_type == "rotation" else 2 * N
    torch.manual_seed(0)
    B = G
    x = _smooth(B, C, H, W, 0)
    gy = torch.randn(B, C, H, W)
    gidx = torch.arange(B) % G
    ang = io.group_angles(N)
    rot0 = (torch.cat([ang, ang]) if G > N else ang)[gidx]
    ref0 = (gidx >= N).float() if G > N else None

    # oracle
    xr = x.clone().requires_grad_(True)
    rot = rot0.clone().requires_grad_(True)
    ref = ref0.clone().requires_grad_(True) if ref0 is not None else None
    (io.canonicalize_images(xr, rot, ref, shape) * gy).sum().backward()

    # HIP
    pad = math.ceil(W * 0.5)
    theta, flags = device_tables("canonicalize", N, G > N, (H + 2 * pad, W + 2 * pad), dev)
    xd = x.to(dev).requires_grad_(True)
    rotd = rot0.to(dev).requires_grad_(True)
    refd = ref0.to(dev).requires_grad_(True) if ref0 is not None else None
    y = _CanonTransformFn.apply(xd, rotd, refd, gidx.to(dev, torch.int32), theta, flags, pad, N)
    (y * gy.to(dev)).sum().backward()

    assert _rel(xd.grad, xr.grad) < 1e-4
    if ref is not None:
        assert _rel(refd.grad, ref.grad) < 1e-4
    _check_angle_grad(rotd.grad, rot.grad, rot0)


@pytest.mark.parametrize("group_type,N,rep", [("rotation", 8, "scalar"), ("rotation", 4, "regular"),
                                              ("roto-reflection", 4, "regular"), ("roto-reflection", 4, "scalar")])
def test_invert_gradients(dev, group_type, N, rep):
    from equiadapt_amd.images.utils import get_action_on_image_features

    G = N if group_type == "rotation" else 2 * N
    torch.manual_seed(1)
    B, H, W = G + 1, 36, 44
    C = 2 * G if rep == "regular" else 3
    f = _smooth(B, C, H, W, 1)
    g = torch.randn(B, C, H, W)
    gidx = (torch.arange(B) * 3 + 1) % G
    ang = io.group_angles(N)
    rot0 = (torch.cat([ang, ang]) if G > N else ang)[gidx]
    ref0 = (gidx >= N).float() if G > N else None

    fr = f.clone().requires_grad_(True)
    rot = rot0.clone().requires_grad_(True)
    ref = ref0.clone().requires_grad_(True) if ref0 is not None else None
    (io.invert_action(fr, rot, ref, N, G, rep) * g).sum().backward()

    fd = f.to(dev).requires_grad_(True)
    el = {"rotation": rot0.to(dev).requires_grad_(True), "group_index": gidx.to(dev, torch.int32)}
    if ref0 is not None:
        el["reflection"] = ref0.to(dev).requires_grad_(True)
    out = get_action_on_image_features(fd, {"num_rotations": N, "num_group": G}, el, rep)
    (out * g.to(dev)).sum().backward()

    assert _rel(fd.grad, fr.grad) < 1e-4
    if ref is not None:
        assert _rel(el["reflection"].grad, ref.grad) < 1e-4
    # the regular-representation roll is piecewise constant in the angle (shift.long()): no gradient through it
    _check_angle_grad(el["rotation"].grad, rot.grad, rot0)


def test_task_loss_reaches_the_canonicalization_network(dev):
    """End to end in train mode: a loss on the canonicalized image produces gradients on the canon net's weights
    (through d/d rotation -> straight-through one-hot -> softmax -> network), as in the reference."""
    import types

    import equiadapt_amd as ea

    torch.manual_seed(2)
    net = ea.CustomEquivariantNetwork((3, 32, 32), 4, 5, "roto-reflection", 4, 2, device="cpu")
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=0.8, resize_shape=32)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 48, 48)).to(dev).train()
    x = torch.randn(6, 3, 48, 48, device=dev)
    y = can(x)
    # invert in train mode is differentiable w.r.t. the prediction output (and the group element) too
    f = torch.randn(6, 8, 48, 48, device=dev, requires_grad=True)
    inv = can.invert_canonicalization(f)
    loss = (y * torch.randn_like(y)).sum() + can.get_prior_regularization_loss() + (inv * torch.randn_like(inv)).sum()
    loss.backward()
    grads = [p.grad for p in can.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    assert sum(g.abs().sum().item() for g in grads) > 0
    assert f.grad is not None and torch.isfinite(f.grad).all() and f.grad.abs().sum().item() > 0


def test_training_steps_on_device(dev):
    """A few optimisation steps through the HIP forward/backward (reference step semantics, training.py): the prior loss
    falls, both the canonicalization network and the prediction network receive updates."""
    import types

    import equiadapt_amd as ea
    from equiadapt_amd import training as tr

    torch.manual_seed(3)
    net = ea.CustomEquivariantNetwork((3, 24, 24), 4, 5, "rotation", 4, 2, device="cpu")
    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=0.8, resize_shape=24)
    can = ea.GroupEquivariantImageCanonicalization(net, hp, (3, 32, 32))
    pred = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, 2, 1), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(1),
                               torch.nn.Flatten(), torch.nn.Linear(8, 5))
    model = tr.CanonicalizedClassifier(can, pred, tr.LossWeights(task_weight=1.0, prior_weight=10.0)).to(dev).train()
    opt, _ = tr.configure_optimizer(model, 1e-2, 1e-2, kind="adamw")
    before = [p.detach().clone() for p in model.parameters()]
    g = torch.Generator().manual_seed(4)
    x = torch.randn(32, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, 5, (32,), generator=g).to(dev)
    priors = []
    for _ in range(12):
        out = tr.train_step(model, opt, x, y)
        priors.append(out["prior_loss"].item())
        assert torch.isfinite(out["loss"]).item()
    assert priors[-1] < priors[0]
    moved = [not torch.equal(a, b.detach()) for a, b in zip(before, model.parameters())]
    assert all(moved)
    # eval mode afterwards: the inference fast paths (window sums / fused kernels) agree with the training-mode modules
    model.eval()
    with torch.no_grad():
        a_fast = can.canonicalization_network(can.transformations_before_canonicalization_network_forward(x))
    with torch.enable_grad():
        a_slow = can.canonicalization_network(can.transformations_before_canonicalization_network_forward(x)).detach()
    assert torch.allclose(a_fast, a_slow, atol=1e-5, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [32, 30, 96])
def test_escnn_training_fast_path_matches_module_path(dev, size, monkeypatch):
    """ESCNNEquivariantNetwork in train(): the Winograd / window-sum autograd path vs the plain module sequence
    (F.conv2d + BatchNorm3d + group_pool through autograd) with the same weights: activations, every parameter gradient, the
    input gradient and the batch-norm running statistics.  size 32 exercises F(4x4,5x5), size 30 F(2x2,5x5), size 96 (batch 9:
    36 tiles) the FFT convolution in the forward pass with the Winograd kernels in the backward pass."""
    import copy

    import equiadapt_amd as ea

    torch.manual_seed(61)
    net = ea.ESCNNEquivariantNetwork((3, size, size), out_channels=8, kernel_size=5, group_type="rotation", num_rotations=8,
                                     num_layers=3).to(dev)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0                                    # same function on both sides
        if hasattr(m, "bias") and isinstance(getattr(m, "bias"), torch.nn.Parameter):
            torch.nn.init.normal_(m.bias, std=0.1)
    # size 96: the reference is the module sequence in fp64 (the fp32 module path goes through whatever convolution kernels the
    # framework's auto-tuner picks on the box -- on some boxes kernels that are themselves 1e-2 off in the input gradient; the
    # FFT path is within 2e-6 of fp64, closer than any fp32 module path)
    ref = copy.deepcopy(net).double() if size == 96 else copy.deepcopy(net)
    rdt = torch.float64 if size == 96 else torch.float32
    net.train()
    ref.train()
    nb = 9 if size == 96 else 6
    x = torch.randn(nb, 3, size, size, device=dev)
    x1, x2 = x.clone().requires_grad_(True), x.clone().to(rdt).requires_grad_(True)
    w = torch.randn(nb, 8, device=dev)
    if size == 96:
        from equiadapt_amd.images.canonicalization_networks import fftconv

        assert fftconv.TRAIN_FORWARD and fftconv.applicable(
            torch.zeros(nb, 64, 92, 92, device=dev).contiguous(memory_format=torch.channels_last), 64, 64)

    assert net._training_fast_path_ok(x1)
    a1 = net(x1)
    monkeypatch.setenv("EQA_TRAIN_FAST", "0")
    a2 = ref(x2)
    monkeypatch.delenv("EQA_TRAIN_FAST")
    assert a1.shape == a2.shape == (nb, 8)
    scale = a2.abs().max().item()
    assert (a1 - a2).abs().max().item() <= 2e-5 * max(scale, 1.0), (a1 - a2).abs().max().item()
    (a1 * w).sum().backward()
    (a2 * w.to(rdt)).sum().backward()
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), ref.named_parameters()):
        assert n1 == n2
        assert p1.grad is not None, n1                    # DistributedDataParallel needs a gradient for every parameter
        g = p2.grad.abs().max().item()
        if g <= 1e-5:
            # a convolution bias in front of a batch-norm cancels in the normalised output: exactly zero gradient on the
            # fast path, rounding noise on the module path
            assert n1.endswith("bias") and p1.grad.abs().max().item() == 0.0, n1
            continue
        rtol = 5e-5 if size == 96 else 3e-3      # measured against fp64: FFT path 1-2e-6, Winograd path up to 8e-4
        assert p1.grad is not None and (p1.grad - p2.grad).abs().max().item() <= rtol * g, (n1, (p1.grad - p2.grad).abs().max().item(), g)
    gx = x2.grad.abs().max().item()
    assert (x1.grad - x2.grad).abs().max().item() <= (5e-5 if size == 96 else 3e-3) * gx, ((x1.grad - x2.grad).abs().max().item(), gx)
    for (n1, b1), (n2, b2) in zip(net.named_buffers(), ref.named_buffers()):
        if "running" in n1 or "num_batches" in n1:
            assert torch.allclose(b1.float(), b2.float(), rtol=1e-4, atol=1e-5), n1
    # eval mode with autograd enabled: running statistics, same answer as the module path
    net.eval()
    ref.eval()
    b1 = net(x)
    monkeypatch.setenv("EQA_TRAIN_FAST", "0")
    b2 = ref(x.to(rdt))
    assert (b1 - b2).abs().max().item() <= 2e-5 * max(b2.abs().max().item(), 1.0)
    # frozen batch-norm fine-tuning (running statistics under autograd): the hidden convolution biases do NOT cancel any more
    # -- out = scale * (h + conv_bias - running_mean) + beta -- and must receive the gradient the module path gives them
    monkeypatch.delenv("EQA_TRAIN_FAST")
    net.zero_grad()
    ref.zero_grad()
    (b1 * w).sum().backward()
    (b2 * w.to(rdt)).sum().backward()
    worst = {}
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), ref.named_parameters()):
        g = p2.grad.abs().max().item()
        assert g > 1e-6, n1
        worst[n1] = (p1.grad - p2.grad).abs().max().item() / g
    # without the batch-norm's renormalisation the gradients of the early layers are differences of large terms: measured
    # 6e-4 (FFT path vs fp64) and 4e-3 (Winograd path vs the fp32 module path) on the lifting layer's filters
    rtol = 2e-3 if size == 96 else 8e-3
    assert all(v <= rtol for v in worst.values()), ("frozen bn", worst)
    assert max(v for k, v in worst.items() if k.endswith("bias")) <= rtol


@pytest.mark.gpu
def test_fused_bn_relu_dropout_block(dev):
    """InnerBnReluDropout (eqa_bn_*): against the op-by-op block without dropout (values, d/dh, d/dgamma, d/dbeta), and the
    dropout mask itself: keep rate, 1/(1-p) scaling, reproducible from torch's seed, same mask in the backward."""
    from equiadapt_amd.images.canonicalization_networks.escnn_networks import ESCNNEquivariantNetwork, InnerBnReluDropout, _InnerBatchNorm

    torch.manual_seed(71)
    E, Fd, B, H, W = 8, 6, 5, 9, 11
    h = (torch.randn(B, Fd * E, H, W, device=dev) * 2 + 0.5).contiguous(memory_format=torch.channels_last)
    for training in (True, False):
        bn1, bn2 = _InnerBatchNorm(Fd, momentum=0.9).to(dev), _InnerBatchNorm(Fd, momentum=0.9).to(dev)
        for bn in (bn1, bn2):
            with torch.no_grad():
                bn.weight.copy_(torch.linspace(0.5, 1.5, Fd)); bn.bias.copy_(torch.linspace(-0.3, 0.3, Fd))
                bn.running_mean.copy_(torch.linspace(-0.2, 0.4, Fd)); bn.running_var.copy_(torch.linspace(0.8, 3.0, Fd))
            bn.train(training)
        cb = torch.linspace(-1, 1, Fd, device=dev)
        h1, h2 = h.clone().requires_grad_(True), h.clone().requires_grad_(True)
        y1 = InnerBnReluDropout.apply(h1, bn1.weight, bn1.bias, bn1, E, cb, 0.5, False)
        y2 = torch.relu(ESCNNEquivariantNetwork._inner_bn(h2, bn2, E, cb))
        assert torch.allclose(y1, y2, atol=2e-5)
        w = torch.randn_like(h)
        (y1 * w).sum().backward()
        (y2 * w).sum().backward()
        assert torch.allclose(h1.grad, h2.grad, atol=2e-4 * h2.grad.abs().max().item())
        assert torch.allclose(bn1.weight.grad, bn2.weight.grad, rtol=1e-4, atol=1e-3)
        assert torch.allclose(bn1.bias.grad, bn2.bias.grad, rtol=1e-4, atol=1e-3)
        assert torch.allclose(bn1.running_mean, bn2.running_mean, atol=1e-5) and torch.allclose(bn1.running_var, bn2.running_var, rtol=1e-5)
    # dropout
    bn = _InnerBatchNorm(Fd).to(dev).train()
    big = torch.randn(16, Fd * E, 32, 32, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    torch.manual_seed(5)
    ya = InnerBnReluDropout.apply(big, bn.weight, bn.bias, bn, E, None, 0.5, True)
    torch.manual_seed(5)
    yb = InnerBnReluDropout.apply(big, bn.weight, bn.bias, bn, E, None, 0.5, True)
    yc = InnerBnReluDropout.apply(big, bn.weight, bn.bias, bn, E, None, 0.5, True)
    y0 = InnerBnReluDropout.apply(big, bn.weight, bn.bias, bn, E, None, 0.0, True)
    assert torch.equal(ya, yb) and not torch.equal(ya, yc)
    pos = y0 > 0
    kept = (ya > 0) & pos
    rate = kept.sum().item() / pos.sum().item()
    assert abs(rate - 0.5) < 5e-3, rate
    assert torch.allclose(ya[kept], 2.0 * y0[kept], rtol=1e-6)
    assert (ya[~pos] == 0).all()
    # per-channel keep rate is uniform too (the hash must not correlate with the channel index)
    per_c = kept.sum(dim=(0, 2, 3)).float() / pos.sum(dim=(0, 2, 3)).float()
    assert (per_c - 0.5).abs().max().item() < 0.03
    g, = torch.autograd.grad(ya.sum(), big)
    g0, = torch.autograd.grad((y0 * kept * 2.0).sum(), big)      # same function with the mask written out
    assert torch.allclose(g, g0, atol=1e-4 * g0.abs().max().item())
    # the backward passes with y given and with "kept and positive" recomputed (y = NULL) are bit-identical
    from equiadapt_amd import _lib, ops
    lib = _lib.load()
    C, npix = Fd * E, 16 * 32 * 32
    xh = big.detach()
    scale = (torch.rand(C, device=dev) + 0.5) * torch.where(torch.arange(C, device=dev) % 3 == 0, -1.0, 1.0)
    shift, mean, rstd = torch.randn(C, device=dev) * 0.3, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
    a, b, d = torch.randn(C, device=dev), torch.randn(C, device=dev), torch.randn(C, device=dev)
    gy = torch.randn_like(xh)
    st = ops._stream()
    for p_drop, seed in ((0.5, 1234567), (0.0, 0), (0.25, 7)):
        yy = torch.empty_like(xh)
        _lib.check(lib.eqa_bn_relu_dropout_nhwc(xh.data_ptr(), scale.data_ptr(), shift.data_ptr(), yy.data_ptr(), npix, C, p_drop, seed, st), "fwd")
        nblk = lib.eqa_bn_partial_blocks(npix)
        outs = []
        for yptr in (yy.data_ptr(), None):
            part = torch.zeros((nblk, C, 2), dtype=torch.float64, device=dev)
            dx = torch.empty_like(xh)
            _lib.check(lib.eqa_bn_bwd_reduce_nhwc(gy.data_ptr(), yptr, xh.data_ptr(), mean.data_ptr(), rstd.data_ptr(), p_drop, part.data_ptr(),
                                                  npix, C, scale.data_ptr(), shift.data_ptr(), seed, st), "reduce")
            _lib.check(lib.eqa_bn_bwd_apply_nhwc(gy.data_ptr(), yptr, xh.data_ptr(), mean.data_ptr(), rstd.data_ptr(), a.data_ptr(), b.data_ptr(),
                                                 d.data_ptr(), p_drop, dx.data_ptr(), npix, C, scale.data_ptr(), shift.data_ptr(), seed, st), "apply")
            outs.append((part, dx))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (p_drop, seed)


@pytest.mark.gpu
def test_input_gradient_gather_equals_atomic_scatter(dev):
    """Un-padded input gradient: the deterministic gather (default) against the atomic scatter (eqa_set_option(0, 1)), for
    rotations incl. 45 degrees, output flips, regular-representation channel maps and non-square, odd sizes; and it is
    bit-reproducible run to run."""
    from equiadapt_amd import _lib, ops
    from equiadapt_amd.images.utils import device_tables

    lib = _lib.load()
    torch.manual_seed(91)
    for (N, refl, C, H, W, regular) in [(8, False, 3, 37, 37, False), (4, True, 8, 40, 40, True), (8, True, 16, 33, 33, True),
                                        (8, False, 2, 224, 224, False)]:
        G = 2 * N if refl else N
        th, fl, cm = device_tables("invert", N, refl, (H, W), dev)
        B = 9
        src = torch.randn(B, C, H, W, device=dev)
        gy = torch.randn(B, C, H, W, device=dev)
        gidx = torch.randint(0, G, (B,), device=dev, dtype=torch.int32)
        cmap = cm if regular else None
        g1, _ = ops.group_action_bwd(src, gy, gidx, th, fl, cmap, 0, (0, 0), True, False)
        g1b, _ = ops.group_action_bwd(src, gy, gidx, th, fl, cmap, 0, (0, 0), True, False)
        lib.eqa_set_option(0, 1)
        try:
            g2, _ = ops.group_action_bwd(src, gy, gidx, th, fl, cmap, 0, (0, 0), True, False)
        finally:
            lib.eqa_set_option(0, 0)
        assert torch.equal(g1, g1b)
        # the two kernels evaluate the sample coordinate with different FMA contraction: at 45 degrees and |coordinate| ~ 100
        # that is ~1e-5 px, i.e. ~1e-5 of a bilinear weight (both are 7e-5 from the fp64 adjoint at 224 x 224)
        assert (g1 - g2).abs().max().item() <= 1e-4, (N, refl, C, H, W, (g1 - g2).abs().max().item())
        # adjoint identity <T x, g> == <x, T^* g>
        y = ops.group_action(src, gidx, th, fl, cmap, 0, (H, W), (0, 0))
        lhs, rhs = (y * gy).sum().item(), (src * g1).sum().item()
        assert abs(lhs - rhs) <= 1e-3 * max(abs(lhs), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [20, 16, 27, 3])
def test_vnsmall_training_fast_path_matches_op_path(dev, monkeypatch, k):
    """VNSmall in train(): the recompute kernels of the first block (eqa_vn_*) vs the op-by-op path with the same weights:
    output vectors, every parameter gradient, the batch-norm running statistics; and the kNN kernel vs torch.topk.
    k = 20: five edges per lane; 16 / 3: partly idle lanes of the k <= 20 instantiation; 27: the k <= 32 instantiation."""
    import copy
    import types

    import equiadapt_amd as ea
    from equiadapt_amd import _lib
    from equiadapt_amd.pointcloud.canonicalization_networks.equivariant_networks import knn

    torch.manual_seed(101)
    hp = types.SimpleNamespace(n_knn=k, pooling="mean")
    net = ea.VNSmall(hp).to(dev)
    net.dropout.p = 0.0
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    ref = copy.deepcopy(net)
    x = torch.randn(5, 3, 200, device=dev)
    # kNN: same neighbour SETS as topk on the reference's score matrix (order within the list is irrelevant for the mean)
    lib = _lib.load()
    idx = torch.empty(5, 200, k, dtype=torch.int32, device=dev)
    assert lib.eqa_vn_knn(x.data_ptr(), idx.data_ptr(), 5, 200, k, None) == 0
    want = knn(x, k)
    assert torch.equal(idx.long().sort(-1).values, want.sort(-1).values)
    # the block itself, with a FIXED upstream gradient, against the op-by-op block evaluated in fp64
    from equiadapt_amd.pointcloud.canonicalization_networks.equivariant_networks import ConvPosMeanPool, get_graph_feature_cross

    g_up = torch.randn(5, 21, 3, 200, device=dev)
    for training in (True, False):
        fast, op64 = copy.deepcopy(net).train(training), copy.deepcopy(net).double().train(training)
        cp = fast.conv_pos
        o1 = ConvPosMeanPool.apply(x, cp.map_to_feat.weight, cp.map_to_dir.weight, cp.batchnorm.bn2d.weight, cp.batchnorm.bn2d.bias,
                                   cp.batchnorm.bn2d, k)
        (o1 * g_up).sum().backward()
        o2 = op64.conv_pos(get_graph_feature_cross(x.double().unsqueeze(1), k, want)).mean(-1)
        (o2 * g_up.double()).sum().backward()
        assert (o1.double() - o2).abs().max().item() <= 1e-5
        for p1, p2 in zip(fast.conv_pos.parameters(), op64.conv_pos.parameters()):
            assert (p1.grad.double() - p2.grad).abs().max().item() <= 1e-4 * p2.grad.abs().max().item()
        for b1, b2 in zip(fast.conv_pos.buffers(), op64.conv_pos.buffers()):
            assert torch.allclose(b1.double(), b2.double(), rtol=1e-5, atol=1e-7)
    # the whole network.  Its later VN-ReLU gates (<q, d> >= 0) can flip on the 1e-6 differences between the two paths'
    # pooled features; with 1000 points one flipped gate moves a gradient entry by up to ~1 %: hence the loose bound here.
    for training in (True, False):
        net.train(training)
        ref.train(training)
        w = torch.randn(5, 3, 3, device=dev)
        for p in list(net.parameters()) + list(ref.parameters()):
            p.grad = None
        a1 = net(x)
        monkeypatch.setenv("EQA_TRAIN_FAST", "0")
        a2 = ref(x)
        monkeypatch.delenv("EQA_TRAIN_FAST")
        assert torch.allclose(a1, a2, atol=2e-5 * max(a2.abs().max().item(), 1.0))
        (a1 * w).sum().backward()
        (a2 * w).sum().backward()
        for (n1, p1), (n2, p2) in zip(net.named_parameters(), ref.named_parameters()):
            assert (p1.grad is None) == (p2.grad is None), n1
            if p2.grad is None:
                continue
            g = max(p2.grad.abs().max().item(), 1e-6)
            assert (p1.grad - p2.grad).abs().max().item() <= 5e-2 * g, (training, n1, (p1.grad - p2.grad).abs().max().item(), g)
        for (n1, b1), (n2, b2) in zip(net.named_buffers(), ref.named_buffers()):
            assert torch.allclose(b1.float(), b2.float(), rtol=1e-4, atol=1e-6), n1


@pytest.mark.gpu
def test_vnsmall_eval_after_train_forward_sees_the_new_running_statistics(dev, monkeypatch):
    """eval -> train-mode forward WITHOUT an optimizer step (frozen canonicalizer, batch-norm recalibration) -> eval: the
    kernels write the running statistics through raw pointers, and the fused eval kernel folds them from a cache keyed on the
    tensors' versions -- the second eval forward must use the updated statistics, like the op-by-op path and the reference
    (pointcloud/canonicalization_networks/equivariant_networks.py:128-150 with nn.BatchNorm in train mode)."""
    import copy
    import types

    import equiadapt_amd as ea

    torch.manual_seed(7)
    hp = types.SimpleNamespace(n_knn=20, pooling="mean")
    net = ea.VNSmall(hp).to(dev)
    ref = copy.deepcopy(net)
    x = torch.randn(6, 3, 256, device=dev) * 1.7 + 0.3        # statistics well away from the initial (0, 1)
    net.eval()
    with torch.no_grad():
        before = net(x)
    versions = [b._version for b in net.buffers()]
    net.train()
    net(x)                                                    # grad enabled, no backward, no optimizer step
    assert all(b._version > v for b, v in zip(net.buffers(), versions)), "the kernel's writes must bump the buffers' versions"
    net.eval()
    with torch.no_grad():
        after = net(x)
    monkeypatch.setenv("EQA_TRAIN_FAST", "0")
    ref.train()
    ref(x)
    ref.eval()
    with torch.enable_grad():                                 # op-by-op path (the fused eval kernel runs under no_grad only)
        want = ref(x).detach()
    monkeypatch.delenv("EQA_TRAIN_FAST")
    for (n1, b1), (n2, b2) in zip(net.named_buffers(), ref.named_buffers()):
        assert torch.allclose(b1.float(), b2.float(), rtol=1e-4, atol=1e-6), n1
    assert not torch.allclose(before, after, atol=1e-4), "the statistics moved, so must the eval output"
    assert torch.allclose(after, want, atol=2e-5 * max(want.abs().max().item(), 1.0))


@pytest.mark.gpu
def test_padded_input_gradient_frame_gather_equals_atomic_scatter(dev):
    """Edge-padded canonicalize: gather on the padded frame + fold of the pad strips / corners (default) against the atomic
    scatter (eqa_set_option(0, 1)); with reflections, 45-degree elements, odd sizes; bit-reproducible; adjoint identity."""
    import math

    from equiadapt_amd import _lib, ops
    from equiadapt_amd.images.utils import device_tables

    lib = _lib.load()
    torch.manual_seed(93)
    for (N, refl, C, H, W) in [(8, False, 3, 37, 37), (4, True, 2, 40, 40), (8, True, 3, 33, 47), (8, False, 3, 224, 224)]:
        G = 2 * N if refl else N
        pad = math.ceil(W * 0.5)
        th, fl = device_tables("canonicalize", N, refl, (H + 2 * pad, W + 2 * pad), dev)
        B = 7
        src = torch.randn(B, C, H, W, device=dev)
        gy = torch.randn(B, C, H, W, device=dev)
        gidx = torch.randint(0, G, (B,), device=dev, dtype=torch.int32)
        g1, a1 = ops.group_action_bwd(src, gy, gidx, th, fl, None, pad, (pad, pad), True, True)
        g1b, _ = ops.group_action_bwd(src, gy, gidx, th, fl, None, pad, (pad, pad), True, False)
        lib.eqa_set_option(0, 1)
        try:
            g2, a2 = ops.group_action_bwd(src, gy, gidx, th, fl, None, pad, (pad, pad), True, True)
        finally:
            lib.eqa_set_option(0, 0)
        assert torch.equal(g1, g1b)
        assert (g1 - g2).abs().max().item() <= 1e-4 * max(g2.abs().max().item(), 1.0), (N, refl, C, H, W, (g1 - g2).abs().max().item())
        # (two instantiations of the angle kernel: different FMA contraction in sums with cancellation)
        assert torch.allclose(a1, a2, rtol=2e-3, atol=2e-3 * max(a2.abs().max().item(), 1.0))
        y = ops.canon_transform(src, gidx, th, fl, pad)
        lhs, rhs = (y * gy).sum().item(), (src * g1).sum().item()
        assert abs(lhs - rhs) <= 1e-3 * max(abs(lhs), 1.0)


@pytest.mark.gpu
def test_window_sums_backward_expand_matches_index_lookups(dev):
    """eqa_window_sums_bwd_expand_nhwc (class table -> (B,H,W,C) gradient map) vs the two index lookups it replaces, bit
    for bit, and the whole WindowSumsFunction backward vs autograd through an explicit unfold-and-sum in fp64."""
    from equiadapt_amd import _lib, ops
    from equiadapt_amd.images.canonicalization_networks.pooling import WindowSumsFunction

    lib = _lib.load()
    torch.manual_seed(5)
    for (B, H, W, C, k) in [(3, 20, 17, 8, 5), (2, 9, 9, 4, 5), (1, 12, 30, 64, 3), (5, 7, 8, 12, 2)]:
        nb = k - 1
        T = 2 * nb + 1
        table = torch.randn(B, T, T, C, device=dev)
        got = torch.empty(B, H, W, C, device=dev)
        assert lib.eqa_window_sums_bwd_expand_nhwc(table.data_ptr(), got.data_ptr(), B, H, W, C, k, ops._stream()) == 0

        def idx(n):
            i = torch.arange(n, device=dev)
            return torch.where(i < nb, i, torch.where(i >= n - nb, i - (n - nb) + nb + 1, torch.full_like(i, nb)))

        assert torch.equal(got, table[:, idx(H)][:, :, idx(W)]), (B, H, W, C, k)
        x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wS = torch.randn(B, C, k, k, device=dev, dtype=torch.float64)
        (WindowSumsFunction.apply(x, k) * wS).sum().backward()
        x64 = x.detach().double().requires_grad_(True)
        S = torch.stack([torch.stack([x64[:, :, u:u + H - k + 1, v:v + W - k + 1].sum((-1, -2)) for v in range(k)], -1)
                         for u in range(k)], -2)
        (S * wS).sum().backward()
        assert torch.allclose(x.grad.double(), x64.grad, rtol=1e-5, atol=1e-5), (B, H, W, C, k)
    # refused, not approximated: channels not a multiple of 4, map smaller than the two borders
    t = torch.zeros(1, 9, 9, 6, device=dev)
    o = torch.zeros(1, 9, 9, 6, device=dev)
    assert lib.eqa_window_sums_bwd_expand_nhwc(t.data_ptr(), o.data_ptr(), 1, 9, 9, 6, 5, None) == -3
    assert lib.eqa_window_sums_bwd_expand_nhwc(t.data_ptr(), o.data_ptr(), 1, 8, 9, 8, 5, None) == -1


@pytest.mark.gpu
def test_fft_filter_gradient_matches_conv2d_weight(dev):
    """fftconv.filter_grad (spectra of the gradient tiles, one batched GEMM over the tiles, inverse restricted to 5x5) vs
    torch.nn.grad.conv2d_weight in fp64: exact tiling (92 -> 88), partial tiles, channel counts on the fused and the two-pass
    kernels."""
    from equiadapt_amd.images.canonicalization_networks import fftconv

    torch.manual_seed(31)
    for (B, Cin, Cout, H, W) in [(3, 16, 32, 92, 92), (2, 8, 12, 60, 97), (9, 64, 64, 92, 92), (2, 4, 4, 48, 50)]:
        x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(Cout, Cin, 5, 5, device=dev) / (5 * Cin ** 0.5)
        dy = torch.randn(B, Cout, H - 4, W - 4, device=dev).contiguous(memory_format=torch.channels_last)
        keep: list = []
        fftconv.conv5x5(x, fftconv.filter_spectra(w), None, False, keep_V=keep)
        got = fftconv.filter_grad(keep[0], dy, Cin)
        want = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double())
        assert got.shape == want.shape
        assert (got.double() - want).abs().max().item() <= 5e-6 * want.abs().max().item(), (B, Cin, Cout, H, W)


@pytest.mark.gpu
def test_fft_input_gradient_matches_conv2d_input(dev, monkeypatch):
    """fftconv.input_grad (gradient-tile spectra x FFT(filter), inverse with overlap-add of the 48 x 48 blocks) vs
    torch.nn.grad.conv2d_input in fp64, fused and two-pass transforms, exact and partial tiles; and run twice: bit-identical
    (every element is written once, in a fixed order)."""
    from equiadapt_amd.images.canonicalization_networks import fftconv

    torch.manual_seed(32)
    for (B, Cin, Cout, H, W) in [(3, 16, 32, 92, 92), (2, 8, 12, 60, 97), (2, 64, 16, 92, 92), (2, 4, 4, 48, 50), (1, 16, 16, 140, 49)]:
        w = torch.randn(Cout, Cin, 5, 5, device=dev) / (5 * Cout ** 0.5)
        dy = torch.randn(B, Cout, H - 4, W - 4, device=dev).contiguous(memory_format=torch.channels_last)
        got = fftconv.input_grad(dy, w)
        want = torch.nn.grad.conv2d_input((B, Cin, H, W), w.double(), dy.double())
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        assert (got.double() - want).abs().max().item() <= 5e-6 * want.abs().max().item(), (B, Cin, Cout, H, W)
        assert torch.equal(got, fftconv.input_grad(dy, w))


def test_lift_conv_weight_gradient_matches_conv2d_weight(dev):
    """eqa_lift_conv_wgrad_nhwc (fp32 MFMA, per-wave partials added in a fixed order) vs the fp64 convolution-weight-gradient:
    the headline shape (96 -> 92, 3 -> 256 channels), the other k-step counts the kernel is instantiated for, two channel
    blocks, a 3x3 filter; refusal of shapes outside its envelope; bit-identical repeats.  Tolerance 2e-5 of max|dW|."""
    from equiadapt_amd import ops

    torch.manual_seed(75)
    cases = [(3, 3, 96, 96, 256, 5, 5), (2, 3, 32, 32, 256, 5, 5), (1, 3, 40, 48, 512, 5, 5), (2, 2, 33, 64, 256, 5, 5), (1, 3, 20, 128, 256, 5, 5),
             (2, 3, 30, 30, 256, 3, 3), (1, 1, 12, 96, 256, 5, 5)]
    for (B, Cin, H, W, Cout, kh, kw) in cases:
        x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, Cout, H - kh + 1, W - kw + 1, device=dev).contiguous(memory_format=torch.channels_last)
        assert ops.lift_conv_wgrad_supported(x, Cout, kh, kw), (B, Cin, H, W, Cout, kh, kw)
        got = ops.lift_conv_wgrad_nhwc(x, dy, kh, kw)
        want = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, kh, kw), dy.double())
        assert got.shape == want.shape
        err = (got.double() - want).abs().max().item()
        assert err <= 2e-5 * want.abs().max().item(), (B, Cin, H, W, Cout, kh, kw, err)
        assert torch.equal(got, ops.lift_conv_wgrad_nhwc(x, dy, kh, kw))
    x = torch.randn(2, 3, 96, 96, device=dev).contiguous(memory_format=torch.channels_last)
    assert not ops.lift_conv_wgrad_supported(x, 64, 5, 5)          # channel count not a multiple of 256
    assert not ops.lift_conv_wgrad_supported(x[:, :, :, :95], 256, 5, 5)   # output width not a multiple of 4
    assert not ops.lift_conv_wgrad_supported(x, 256, 7, 7)         # 147 taps
    with pytest.raises(Exception):
        ops.lift_conv_wgrad_nhwc(x, torch.zeros(2, 64, 92, 92, device=dev).contiguous(memory_format=torch.channels_last), 5, 5)


@pytest.mark.gpu
def test_fused_bn_block_wide_channels_partial_last_trip(dev):
    """C = 1536 (D8 with 96 fields): C / 4 = 384 channel quads > 256 threads and not a multiple of 256, so the statistics and
    backward-reduce kernels make two trips over the quads, the second one partial.  Every thread must still reach both
    barriers of both trips (uniform trip count) and the partial trip must be correct."""
    from equiadapt_amd.images.canonicalization_networks.escnn_networks import ESCNNEquivariantNetwork, InnerBnReluDropout, _InnerBatchNorm

    torch.manual_seed(72)
    E, Fd, B, H, W = 16, 96, 3, 7, 5
    h = (torch.randn(B, Fd * E, H, W, device=dev) * 1.5 + 0.25).contiguous(memory_format=torch.channels_last)
    bn1, bn2 = _InnerBatchNorm(Fd, momentum=0.9).to(dev).train(), _InnerBatchNorm(Fd, momentum=0.9).to(dev).train()
    for bn in (bn1, bn2):
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, Fd)); bn.bias.copy_(torch.linspace(-0.3, 0.3, Fd))
    h1, h2 = h.clone().requires_grad_(True), h.clone().requires_grad_(True)
    y1 = InnerBnReluDropout.apply(h1, bn1.weight, bn1.bias, bn1, E, None, 0.5, False)
    y2 = torch.relu(ESCNNEquivariantNetwork._inner_bn(h2, bn2, E, None))
    assert torch.allclose(y1, y2, atol=2e-5)
    w = torch.randn_like(h)
    (y1 * w).sum().backward()
    (y2 * w).sum().backward()
    assert torch.allclose(h1.grad, h2.grad, atol=2e-4 * h2.grad.abs().max().item())
    assert torch.allclose(bn1.weight.grad, bn2.weight.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(bn1.bias.grad, bn2.bias.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(bn1.running_mean, bn2.running_mean, atol=1e-5) and torch.allclose(bn1.running_var, bn2.running_var, rtol=1e-5)


@pytest.mark.parametrize("B,N,p", [(4, 256, 0.0), (3, 100, 0.0), (2, 1024, 0.5)])
def test_vnsmall_training_tail_matches_op_path(dev, B, N, p):
    """VNSmall's tail in train() (conv1 -> bn1 -> conv2 -> dropout -> mean; eqa_vn_tail_pass) against the same modules op by op in
    fp64 on the SAME pooled features and the same dropout mask: output, the gradient w.r.t. the pooled features, every parameter
    gradient, running statistics and num_batches_tracked.  N = 100 leaves idle lanes in the last block."""
    import copy
    import types

    import equiadapt_amd as ea
    from equiadapt_amd.pointcloud.canonicalization_networks.equivariant_networks import TailMean

    torch.manual_seed(7 + N)
    net = ea.VNSmall(types.SimpleNamespace(n_knn=20, pooling="mean")).to(dev).train()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
                m.running_mean.uniform_(-0.1, 0.1)
        # conv1's batch-normalised norms gamma * nhat + beta must stay away from 0: bn1 takes the norm of conv1's output, whose
        # gradient is ~1 / |y1|; with beta around 0 a handful of the 20 k vectors land at |y1| ~ 1e-5 of the median and fp32 (this
        # path and the op-by-op one alike, measured 2e-3 vs fp64) cannot resolve their gradients
        net.conv1.batchnorm.bn1d.bias.uniform_(2.5, 3.5)
    op64 = copy.deepcopy(net).double().train()
    pooled = torch.randn(B, 21, 3, N, device=dev, requires_grad=True)
    pooled64 = pooled.detach().double().requires_grad_(True)
    mask = None
    if p > 0:
        mask = torch.nn.functional.dropout(torch.ones(B, 4, 3, N, device=dev), p, True)
    g_up = torch.randn(B, 3, 3, device=dev)
    bns = (net.conv1.batchnorm.bn1d, net.bn1.bn1d, net.conv2.batchnorm.bn1d)
    y = TailMean.apply(pooled, net.conv1.map_to_feat.weight, net.conv1.map_to_dir.weight, net.conv2.map_to_feat.weight,
                       net.conv2.map_to_dir.weight, bns[0].weight, bns[0].bias, bns[1].weight, bns[1].bias, bns[2].weight, bns[2].bias,
                       bns, mask)[:, :3]
    (y * g_up).sum().backward()
    o = op64.conv2(op64.bn1(op64.conv1(pooled64)))
    if mask is not None:
        o = o * mask.double()
    y64 = o.mean(-1)[:, :3]
    (y64 * g_up.double()).sum().backward()
    assert (y.double() - y64).abs().max().item() <= 2e-6 * max(1.0, y64.abs().max().item())
    scale = pooled64.grad.abs().max().item()
    assert (pooled.grad.double() - pooled64.grad).abs().max().item() <= 2e-4 * scale
    tail = lambda n: [q for k, q in n.named_parameters() if k.split(".")[0] in ("conv1", "bn1", "conv2")]  # noqa: E731
    for (k, p1), p2 in zip([(k, q) for k, q in net.named_parameters() if k.split(".")[0] in ("conv1", "bn1", "conv2")], tail(op64)):
        assert p1.grad is not None, k
        assert (p1.grad.double() - p2.grad).abs().max().item() <= 2e-4 * max(p2.grad.abs().max().item(), 1e-3), k
    for (k, b1), b2 in zip(net.named_buffers(), op64.buffers()):
        if k.split(".")[0] in ("conv1", "bn1", "conv2"):
            if b1.dtype.is_floating_point:
                assert torch.allclose(b1.double(), b2, rtol=1e-5, atol=1e-7), k
            else:
                assert int(b1) == int(b2) == 1, k


def test_gram_schmidt_backward_matches_autograd_of_the_oracle(dev):
    """eqa_gram_schmidt_bwd (analytic) against autograd through the oracle's restatement of common/utils.py:22-51 in fp64."""
    from oracle import pointcloud_ops as po
    from equiadapt_amd.common.utils import gram_schmidt

    torch.manual_seed(31)
    v = torch.randn(257, 3, 3)
    g = torch.randn(257, 3, 3)
    v64 = v.double().requires_grad_(True)
    (po.gram_schmidt(v64) * g.double()).sum().backward()
    vd = v.to(dev).requires_grad_(True)
    (gram_schmidt(vd) * g.to(dev)).sum().backward()
    # nearly collinear inputs make the derivative ~1 / sin^2(angle) and fp32 loses it (any fp32 evaluation does): judge the
    # samples whose second and third vectors keep at least 0.2 of their length after the projections
    with torch.no_grad():
        e = po.gram_schmidt(v64)
        u2 = v64[:, 1] - (v64[:, 1] * e[:, 0]).sum(1, keepdim=True) * e[:, 0]
        u3 = (v64[:, 2] * e[:, 2]).sum(1).abs()
        ok = (u2.norm(dim=1) / v64[:, 1].norm(dim=1) > 0.2) & (u3 / v64[:, 2].norm(dim=1) > 0.2)
    assert int(ok.sum()) >= 120
    scale = v64.grad.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-3)
    rel = (vd.grad.cpu().double() - v64.grad).abs() / scale
    assert rel[ok].max().item() <= 2e-4
    assert rel.median().item() <= 1e-5
    assert vd.grad.shape == (257, 3, 3)
    empty = torch.empty(0, 3, 3, device=dev, requires_grad=True)
    gram_schmidt(empty).sum().backward()
    assert empty.grad.shape == (0, 3, 3)


@pytest.mark.parametrize("M,Cin,Cout", [(37, 64, 128), (64, 128, 64), (130, 256, 256)])
def test_fft_wgrad3m_matches_the_real_product(dev, M, Cin, Cout):
    """eqa_fft48k5_wgrad3m (filter-gradient contraction over the tiles, 3-multiplication form on the fp32 MFMA) against the real
    [2Cin x M].[M x 2Cout] product in fp64: Dr = (re,re) + (im,im), Di = (im,re) - (re,im) -- the two numbers
    eqa_fft48k5_filter_grad takes from every channel pair -- to 2e-6 of the largest entry; the padding row of the spectra (tile
    pitch M | 1) is poisoned with NaN to show it is never read; and the filter gradient computed from it
    (eqa_fft48k5_filter_grad3m) equals the library path's."""
    from equiadapt_amd import _lib
    from equiadapt_amd.images.canonicalization_networks import fftconv

    lib = _lib.load()
    torch.manual_seed(M)
    F = fftconv.F
    pitch = M | 1
    V = torch.randn(F, pitch, 2 * Cin, device=dev)
    G = torch.randn(F, pitch, 2 * Cout, device=dev)
    if pitch > M:
        V[:, M:] = float("nan")
        G[:, M:] = float("nan")
    D = torch.full((F, Cin, 2, Cout), float("nan"), device=dev)
    assert lib.eqa_fft48k5_wgrad3m_supported(Cin, Cout) == 1
    _lib.check(lib.eqa_fft48k5_wgrad3m(V.data_ptr(), G.data_ptr(), D.data_ptr(), M, Cin, Cout, None), "eqa_fft48k5_wgrad3m")
    fsel = torch.tensor([0, 1, 7, 500, 1103, 1104, 1153], device=dev)
    ref = torch.bmm(V[fsel, :M].double().transpose(1, 2), G[fsel, :M].double())          # (7, 2Cin, 2Cout)

    def quad(T, a, b):   # (re/im of ci, re/im of co) quadrants of the [Re x 16 | Im x 16] grouped layout
        T = T.reshape(T.shape[0], Cin // 16, 2, 16, Cout // 16, 2, 16)
        return T[:, :, a, :, :, b, :]
    want_r, want_i = quad(ref, 0, 0) + quad(ref, 1, 1), quad(ref, 1, 0) - quad(ref, 0, 1)
    assert torch.isfinite(D).all()
    got = D[fsel].double()                                          # (7, Cin, 2, Cout), plain channel order
    got_r = got[:, :, 0].reshape(-1, Cin // 16, 16, Cout // 16, 16)
    got_i = got[:, :, 1].reshape(-1, Cin // 16, 16, Cout // 16, 16)
    scale = max(want_r.abs().max().item(), want_i.abs().max().item())
    assert (got_r - want_r).abs().max().item() <= 2e-6 * scale
    assert (got_i - want_i).abs().max().item() <= 4e-6 * scale      # a difference of three products
    # through eqa_fft48k5_filter_grad: the same filter gradient as from the library's 4-product form
    db3 = torch.empty(Cout, Cin, 5, 5, device=dev)
    db4 = torch.empty(Cout, Cin, 5, 5, device=dev)
    D4 = torch.bmm(V[:, :M].transpose(1, 2), G[:, :M]).contiguous()
    _lib.check(lib.eqa_fft48k5_filter_grad3m(D.data_ptr(), db3.data_ptr(), Cout, Cin, None), "eqa_fft48k5_filter_grad3m")
    _lib.check(lib.eqa_fft48k5_filter_grad(D4.data_ptr(), db4.data_ptr(), Cout, Cin, None), "eqa_fft48k5_filter_grad")
    assert (db3 - db4).abs().max().item() <= 5e-6 * db4.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("group_type,N,size", [("rotation", 8, 96), ("roto-reflection", 4, 40), ("rotation", 4, 30)])
def test_fused_last_block_into_window_sums_is_bit_identical(dev, group_type, N, size, monkeypatch):
    """The last hidden block consumed by the window sums without being written (InnerBnReluDropoutWindowSums: affine + ReLU +
    dropout mask applied while eqa_window_sums_nhwc_act loads h; backward from the (2k-1)^2 class table) against the two-stage form
    (InnerBnReluDropout writes the block's output, WindowSumsFunction reads it; its backward expands the table to a map): same
    statistics, same mask, same per-element arithmetic and summation order -- activations, every parameter gradient and the
    running statistics must agree bit for bit, with dropout ACTIVE (p = 0.5, seeds from torch's CPU generator)."""
    import copy

    import equiadapt_amd as ea

    torch.manual_seed(5)
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)    # the framework's filter-gradient solvers (small shapes) must not use atomics
    net = ea.ESCNNEquivariantNetwork((3, size, size), 8, 5, group_type, N, 3).to(dev).train()
    ref = copy.deepcopy(net)
    ref2 = copy.deepcopy(net)
    x = torch.randn(6, 3, size, size, device=dev)
    w = torch.randn(6, net.num_group_elements, device=dev)
    outs = []
    for model, fused in ((net, "1"), (ref, "0"), (ref2, "0")):
        monkeypatch.setenv("EQA_TRAIN_FUSED_TAIL", fused)
        torch.manual_seed(99)                       # the dropout seeds
        a = model(x)
        (a * w).sum().backward()
        outs.append(a.detach())
    monkeypatch.delenv("EQA_TRAIN_FUSED_TAIL")
    assert torch.equal(outs[0], outs[1])
    for (n1, p1), (n2, p2), (n3, p3) in zip(net.named_parameters(), ref.named_parameters(), ref2.named_parameters()):
        assert (p1.grad is None) == (p2.grad is None), n1
        if p1.grad is not None:
            assert torch.equal(p2.grad, p3.grad), f"{n1}: the two-stage form is not reproducible run to run"
            assert torch.equal(p1.grad, p2.grad), (n1, (p1.grad - p2.grad).abs().max().item(), p2.grad.abs().max().item())
    for (n1, b1), (n2, b2) in zip(net.named_buffers(), ref.named_buffers()):
        assert torch.equal(b1, b2), n1


@pytest.mark.parametrize("B,HW,k,cout", [(5, (96, 96), 5, 256), (3, (68, 100), 5, 128), (4, (40, 51), 3, 64), (2, (36, 36), 5, 192)])
def test_lift_conv_epilogue_statistics_match_the_stats_pass(dev, B, HW, k, cout):
    """eqa_lift_conv_nhwc_stats: the lifting convolution of the training step (escnn_networks.py:60-66 of the reference, feeding the
    InnerBatchNorm of :67-70) with the batch-norm's per-channel sum / sum of squares taken in the kernel's epilogue: the map is
    bit-identical to eqa_lift_conv_nhwc's, the partial rows add up to what eqa_bn_stats_nhwc's pass over the map gives (fp32 running
    sums per lane vs fp32 per 256 pixels: 1e-5 of the sum of squares), including the rows that take the seam pixels of the
    92 -> 3 x 32 tiling out again (OW = 92, 96, 49, 32: 4, 0, 15 and 0 twice-computed columns)."""
    from equiadapt_amd import _lib, ops

    lib = _lib.load()
    torch.manual_seed(B + k)
    H, W = HW
    x = (torch.randn(B, 3, H, W, device=dev) + 0.3).contiguous(memory_format=torch.channels_last)
    bank = torch.randn(cout, 3, k, k, device=dev) / (k * 3 ** 0.5)
    wpk = ops.pack_lift_weights(bank)
    assert ops.lift_conv_stats_supported(x.shape, k, k, cout)
    y, part = ops.lift_conv_nhwc_stats(x, wpk, k, k)
    want_y = ops.lift_conv_nhwc(x, wpk, None, False, k, k)
    assert torch.equal(y, want_y)
    npix = B * (H - k + 1) * (W - k + 1)
    ref = torch.empty((lib.eqa_bn_partial_blocks(npix), cout, 2), dtype=torch.float64, device=dev)
    _lib.check(lib.eqa_bn_stats_nhwc(y.data_ptr(), ref.data_ptr(), npix, cout, None), "eqa_bn_stats_nhwc")
    torch.cuda.synchronize()
    got, want = part.sum(0), ref.sum(0)
    exact = y.permute(0, 2, 3, 1).reshape(-1, cout).double()
    truth = torch.stack([exact.sum(0), (exact * exact).sum(0)], dim=1)
    scale = truth[:, 1].max().item()
    assert (got - truth).abs().max().item() <= 1e-5 * scale, (got - truth).abs().max().item() / scale
    assert (want - truth).abs().max().item() <= 1e-5 * scale
    # shapes without that form: narrow channel counts, rows shorter than a tile
    assert not ops.lift_conv_stats_supported((2, 3, 32, 32), 5, 5, 32)
    assert not ops.lift_conv_stats_supported((2, 3, 40, 30), 5, 5, 64)
    y2, none = ops.lift_conv_nhwc_stats(x[:, :, :, :30].contiguous(memory_format=torch.channels_last), wpk, k, k)
    assert none is None and y2.shape[-1] == 30 - k + 1


def test_training_step_with_epilogue_statistics_matches_the_separate_pass(dev, monkeypatch):
    """The canonicalization network's training forward / backward with the first block's batch statistics taken in the lifting
    convolution's epilogue (default) against the same step with eqa_bn_stats_nhwc's own pass (EQA_TRAIN_EPILOGUE_STATS=0): same
    dropout seeds, so activations and running statistics agree to the rounding of the statistics (1e-5), gradients to 1e-4 of the step's gradient scale.  The
    hidden layer's statistics come from the FFT convolution's inverse transform (eqa_fft48k5_output_stats) the same way."""
    import equiadapt_amd as ea

    def run(flag):
        monkeypatch.setenv("EQA_TRAIN_EPILOGUE_STATS", flag)
        torch.manual_seed(3)
        net = ea.ESCNNEquivariantNetwork((3, 100, 100), 64 // 4, 5, "rotation", 4, 4).to(dev).train()
        x = torch.randn(12, 3, 100, 100, device=dev)
        torch.manual_seed(11)                     # the dropout seeds come from torch's CPU generator
        out = net(x)
        out.square().sum().backward()
        norms = [m for m in net.modules() if hasattr(m, "running_mean") and m.running_mean is not None]
        return out.detach(), [p.grad.clone() for p in net.parameters()], [m.running_var.clone() for m in norms]

    calls = []
    from equiadapt_amd import ops
    orig = ops.lift_conv_nhwc_stats
    monkeypatch.setattr(ops, "lift_conv_nhwc_stats", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    from equiadapt_amd.images.canonicalization_networks import fftconv
    fcalls = []
    forig = fftconv.conv5x5
    monkeypatch.setattr(fftconv, "conv5x5", lambda *a, **k: (fcalls.append(k.get("stats") is not None), forig(*a, **k))[1])
    o1, g1, v1 = run("1")
    assert calls, "the default training step must take the statistics in the convolution's epilogue"
    assert any(fcalls), "... and the hidden layer's in the FFT convolution's inverse transform"
    n = len(calls)
    o0, g0, v0 = run("0")
    assert len(calls) == n
    assert (o1 - o0).abs().max().item() <= 2e-5 * o0.abs().max().item()
    for a, b in zip(v1, v0):
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()
    # Gradients: the late layers agree to 1e-6 of their own size.  The early layers' gradients are what is left after the batch-norm
    # backward passes subtracted the mean and the x-hat-correlated part of an upstream gradient 300x larger (|g| 1e-3 against 0.3-0.6
    # for the last block here), so a 1e-6 difference in the statistics shows as 1e-3 of THEIR size -- measured, and the same for either
    # form against an fp64 evaluation (tools/diag/epilogue_stats_noise.py).  The bound is therefore on the step's gradient scale.
    scale = max(b.abs().max().item() for b in g0)
    for a, b in zip(g1, g0):
        assert (a - b).abs().max().item() <= 1e-4 * scale
    assert (g1[-1] - g0[-1]).abs().max().item() <= 1e-5 * g0[-1].abs().max().item()


@pytest.mark.parametrize("B,HW,cin,cout", [(8, (92, 92), 64, 64), (10, (84, 88), 32, 64), (40, (48, 47), 32, 64)])
def test_fft_conv_output_statistics_match_the_stats_pass(dev, B, HW, cin, cout):
    """eqa_fft48k5_output_stats: the FFT convolution's inverse transform (the hidden regular -> regular layer of
    escnn_networks.py:67-91 in training) also leaves the per-channel sum / sum of squares of its output for the InnerBatchNorm behind
    it: the map is bit-identical to eqa_fft48k5_output's, the partial rows add up to the statistics of the map (full tiles, partial
    border tiles in both directions, a single tile)."""
    from equiadapt_amd.images.canonicalization_networks import fftconv

    torch.manual_seed(B)
    H, W = HW
    x = (torch.randn(B, cin, H, W, device=dev) + 0.2).contiguous(memory_format=torch.channels_last)
    bank = torch.randn(cout, cin, 5, 5, device=dev) / (5 * cin ** 0.5)
    if not fftconv.applicable(x, cin, cout):
        pytest.skip("no FFT path for this shape")
    assert fftconv.output_stats_supported(B, H - 4, W - 4, cout)
    spectra = fftconv.spectra_for(bank)
    stats = []
    y = fftconv.conv5x5(x, spectra, None, False, stats=stats)
    want_y = fftconv.conv5x5(x, spectra, None, False)
    torch.cuda.synchronize()
    assert torch.equal(y, want_y)
    exact = y.permute(0, 2, 3, 1).reshape(-1, cout).double()
    truth = torch.stack([exact.sum(0), (exact * exact).sum(0)], dim=1)
    got = stats[0].sum(0)
    assert (got - truth).abs().max().item() <= 1e-5 * truth[:, 1].max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("group_type,N,C,hw,pad", [("rotation", 8, 3, (224, 224), True), ("roto-reflection", 4, 8, (40, 56), False),
                                                   ("rotation", 8, 5, (33, 47), True), ("rotation", 4, 2, (64, 64), False)])
def test_angle_gradient_staged_kernel_equals_the_direct_gather(dev, group_type, N, C, hw, pad):
    """dL/d(angle) of the group action (what autograd derives through kornia's rotate in discrete_group.py:213 / images/utils.py:57)
    from the LDS-staged kernel (round 4: the forward's window, its four neighbours gathered for the derivative) against round 3's
    direct-gather kernel (eqa_set_option(0, 1)): same per-pixel expression, same order of additions -- equal to the last bits of a
    tile's sum (fma contraction may differ between the two kernels) -- for the edge-padded canonicalizing transform, the un-padded
    invert action with the regular-representation channel map, channel counts that leave a partial last stage, ragged tiles."""
    from equiadapt_amd import _lib, ops
    from equiadapt_amd.images.utils import device_tables

    H, W = hw
    G = N if group_type == "rotation" else 2 * N
    refl = group_type != "rotation"
    torch.manual_seed(C + H)
    B = 2 * G + 3
    x = torch.randn(B, C, H, W, device=dev)
    go = torch.randn(B, C, H, W, device=dev)
    gidx = (torch.arange(B, device=dev) % G).to(torch.int32)
    if pad:
        p = math.ceil(W * 0.5)
        th, fl = device_tables("canonicalize", N, refl, (H + 2 * p, W + 2 * p), dev)
        args = (x, go, gidx, th, fl, None, p, (p, p), False, True)
    else:
        th, fl, cm = device_tables("invert", N, refl, (H, W), dev)
        args = (x, go, gidx, th, fl, cm if C % G == 0 else None, 0, (0, 0), False, True)
    _, staged = ops.group_action_bwd(*args)
    lib = _lib.load()
    lib.eqa_set_option(0, 1)
    try:
        _, direct = ops.group_action_bwd(*args)
    finally:
        lib.eqa_set_option(0, 0)
    scale = direct.abs().max().item()
    assert (staged - direct).abs().max().item() <= 2e-6 * scale, ((staged - direct).abs().max().item(), scale)
