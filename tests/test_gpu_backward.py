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
