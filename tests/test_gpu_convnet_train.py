"""ConvNetwork in TRAINING on the library's own kernels (csrc/convnet_train.hip): filter / data gradients of the
stride-2 convolutions, batch-norm + activation blocks, and the whole training step against the framework's autograd.

Reference: equiadapt/images/canonicalization_networks/custom_nonequivariant_networks.py:8-80.  The reference-generated vectors
(tests/golden/conv_network.pt: train-mode output, running statistics after the step, every parameter gradient) run through the same
path in tests/test_gpu_reference_goldens.py::test_conv_network_on_product_matches_reference_golden.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    from equiadapt_amd import _lib

    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


CASES = [  # (B, Cin, H, W, Cout, K, pad, planar)
    (5, 3, 64, 64, 16, 5, 0, True), (3, 3, 33, 41, 32, 7, 1, True), (4, 1, 20, 20, 16, 3, 0, True), (2, 4, 31, 29, 32, 7, 0, True),
    (5, 16, 30, 30, 16, 5, 0, False), (3, 16, 13, 13, 32, 5, 1, False), (2, 32, 28, 27, 32, 7, 1, False), (3, 32, 17, 19, 64, 3, 0, False),
    (2, 64, 12, 12, 64, 5, 1, False), (7, 16, 61, 61, 16, 7, 0, False),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_s2_filter_and_data_gradients_match_the_framework(dev, case):
    """eqa_conv_s2_wgrad / eqa_conv_s2_dgrad against torch.nn.grad.conv2d_weight / conv2d_input evaluated in fp64 on the CPU (fp32
    products and sums over up to 10^5 pixels: 2e-5 of the gradient's scale).  Every dx element is written (NaN-filled beforehand)."""
    from equiadapt_amd import ops

    B, Cin, H, W, Cout, K, pad, planar = case
    assert ops.conv_s2_train_supported(Cin, Cout, K, pad, planar)
    g = torch.Generator().manual_seed(sum(case[:7]))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) * 0.1
    OH, OW = (H + 2 * pad - K) // 2 + 1, (W + 2 * pad - K) // 2 + 1
    dz = torch.randn(B, Cout, OH, OW, generator=g)
    want_w = torch.nn.grad.conv2d_weight(x.double(), w.shape, dz.double(), stride=2, padding=pad)
    want_x = torch.nn.grad.conv2d_input(x.shape, w.double(), dz.double(), stride=2, padding=pad)
    xd = x.to(dev) if planar else x.permute(0, 2, 3, 1).contiguous().to(dev)
    dzd = dz.permute(0, 2, 3, 1).contiguous().to(dev)
    got_w = ops.conv_s2_wgrad(xd, dzd, K, pad, planar).cpu().double()
    assert (got_w - want_w).abs().max().item() <= 2e-5 * want_w.abs().max().item(), (case, (got_w - want_w).abs().max().item())
    again = ops.conv_s2_wgrad(xd, dzd, K, pad, planar).cpu().double()
    assert torch.equal(got_w, again)                                    # fixed summation order: bit-reproducible
    if not planar:
        got_x = ops.conv_s2_dgrad(dzd, ops.pack_conv_s2_dgrad_weights(w.to(dev)), (H, W), Cin, K, pad)
        assert torch.isfinite(got_x).all()
        got_x = got_x.permute(0, 3, 1, 2).cpu().double()
        assert (got_x - want_x).abs().max().item() <= 2e-5 * want_x.abs().max().item(), (case, (got_x - want_x).abs().max().item())


@pytest.mark.parametrize("act,npix,C,rows", [(0, 2048 * 9, 16, False), (0, 777, 32, False), (1, 300, 1152, True), (1, 64, 8, True), (0, 5, 64, False)])
def test_bn_act_forward_and_backward_match_autograd(dev, act, npix, C, rows):
    """eqa_bn_act_fwd / _bwd_reduce / _bwd_apply (+ eqa_bn_stats_nhwc) against fp64 autograd through
    act(batch_norm(z, batch statistics)) * rowscale."""
    from equiadapt_amd import ops

    g = torch.Generator().manual_seed(act * 100 + C)
    z = torch.randn(npix, C, generator=g) * 1.5 + 0.3
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    rs = ((torch.rand(npix, generator=g) > 0.5).float() * 2.0) if rows else None
    gy = torch.randn(npix, C, generator=g)
    z64 = z.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    h = F.batch_norm(z64, None, None, g64, b64, True, 0.1, 1e-5)
    y64 = F.gelu(h) if act == 0 else F.relu(h)
    if rows:
        y64 = y64 * rs.double()[:, None]
    (y64 * gy.double()).sum().backward()
    zd = z.to(dev)
    mean, var = ops.bn_batch_stats(zd)
    assert (mean.cpu() - z.double().mean(0)).abs().max().item() <= 1e-6
    assert (var.cpu() - z.double().var(0, unbiased=False)).abs().max().item() <= 2e-6
    rstd = torch.rsqrt(var + 1e-5)
    scale = (gamma.to(dev).double() * rstd).float().contiguous()
    shift = (beta.to(dev).double() - mean * gamma.to(dev).double() * rstd).float().contiguous()
    rsd = rs.to(dev) if rows else None
    y = ops.bn_act_fwd(zd, scale, shift, rsd, act)
    assert (y.cpu().double() - y64.detach()).abs().max().item() <= 1e-5
    dz, dgamma, dbeta = ops.bn_act_bwd(gy.to(dev), zd, scale, shift, mean.float().contiguous(), rstd.float().contiguous(), gamma.to(dev), rsd, act)
    sc = z64.grad.abs().max().item()
    assert (dz.cpu().double() - z64.grad).abs().max().item() <= 2e-5 * sc + 1e-6
    assert (dgamma.cpu().double() - g64.grad).abs().max().item() <= 2e-5 * g64.grad.abs().max().item() + 1e-5
    assert (dbeta.cpu().double() - b64.grad).abs().max().item() <= 2e-5 * b64.grad.abs().max().item() + 1e-5


@pytest.mark.parametrize("in_shape,oc,k,L,B", [((3, 64, 64), 16, 5, 3, 48), ((3, 128, 128), 16, 7, 3, 12), ((1, 60, 60), 16, 3, 2, 32), ((3, 200, 200), 16, 3, 6, 16)])
def test_convnetwork_training_step_on_own_kernels_matches_the_framework_path(dev, in_shape, oc, k, L, B, monkeypatch):
    """The same ConvNetwork, the same batch, one training step two ways: the library's kernels (EQA_CONVNET_TRAIN_MODE=hip, default)
    and the framework's modules as constructed (=plain, MIOpen + ATen autograd).  Output, running statistics after the step and
    every parameter gradient agree to fp32 convolution rounding; Dropout1d is active and draws the same per-row mask."""
    import copy

    import equiadapt_amd as ea
    from equiadapt_amd import ops

    torch.manual_seed(5)
    net = ea.ConvNetwork(in_shape, oc, k, L, 32).to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    ref = copy.deepcopy(net)
    x = torch.randn(B, *in_shape, generator=torch.Generator().manual_seed(6)).to(dev)
    up = torch.randn(B, 32, generator=torch.Generator().manual_seed(7)).to(dev)
    assert net._train_hip_applies(x)
    timer = ops.KernelTimer()
    torch.manual_seed(11)
    torch.cuda.manual_seed(11)
    with timer:
        out = net(x)
        (out * up).sum().backward()
    names = timer.summary()
    assert {"conv_s2", "conv_s2_wgrad", "bn_act_fwd", "bn_act_bwd_reduce", "bn_act_bwd_apply"} <= set(names), names
    assert ("conv_s2_dgrad" in names) == (L > 1)
    monkeypatch.setenv("EQA_CONVNET_TRAIN_MODE", "plain")
    torch.manual_seed(11)
    torch.cuda.manual_seed(11)
    want = ref(x)
    (want * up).sum().backward()
    scale = want.abs().max().item()
    assert (out - want).abs().max().item() <= 2e-4 * scale, (out - want).abs().max().item()
    assert (out == 0).all(dim=1).sum().item() == (want == 0).all(dim=1).sum().item()      # the same rows dropped
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if n.startswith("enc_network.") and n.endswith(".bias") and int(n.split(".")[1]) % 3 == 0:
            assert p.grad.abs().max().item() == 0.0                      # a conv bias behind batch statistics: exactly zero here
            wn = dict(ref.named_parameters())[n[:-4] + "weight"].grad.abs().max().item()
            assert q.grad.abs().max().item() <= 1e-3 * wn + 1e-5         # ... rounding noise of a sum that is zero there
            continue
        gs = q.grad.abs().max().item()
        assert (p.grad - q.grad).abs().max().item() <= 2e-3 * gs + 1e-6, (n, (p.grad - q.grad).abs().max().item(), gs)
    for (n, a), (_, b) in zip(net.named_buffers(), ref.named_buffers()):
        assert torch.allclose(a.float(), b.float(), atol=1e-5, rtol=1e-4), n
