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
    (32, 1, 60, 60, 16, 3, 0, True), (32, 16, 29, 29, 16, 3, 0, False), (16, 16, 99, 99, 16, 3, 0, False), (16, 16, 49, 49, 32, 3, 1, False),
    (16, 32, 5, 5, 64, 3, 1, False), (16, 32, 12, 12, 32, 3, 0, False),
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
def test_convnetwork_training_step_on_own_kernels_against_fp64(dev, in_shape, oc, k, L, B, monkeypatch):
    """One training step of the same ConvNetwork on the same batch three ways: the library's kernels (EQA_CONVNET_TRAIN_MODE=hip, the
    default), the framework's modules on the device (=plain: MIOpen + ATen autograd, fp32) and the framework's modules in fp64 on
    the CPU (the truth).  Output, running statistics and every parameter gradient of the library path are within fp32 rounding of the
    truth and no further from it than 1.5 x the framework's own fp32 path (+ 2e-4 of the gradient's scale).  Dropout1d off here (its
    mask is the device generator's); the next test covers it."""
    import copy

    import equiadapt_amd as ea
    from equiadapt_amd import ops

    torch.manual_seed(5)
    net = ea.ConvNetwork(in_shape, oc, k, L, 32)
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    net.final_fc[1].p = 0.0
    truth = copy.deepcopy(net).double().train()
    net = net.to(dev).train()
    ref = copy.deepcopy(net)
    x = torch.randn(B, *in_shape, generator=torch.Generator().manual_seed(6))
    up = torch.randn(B, 32, generator=torch.Generator().manual_seed(7))
    t_out = truth(x.double())
    (t_out * up.double()).sum().backward()
    xd, upd = x.to(dev), up.to(dev)
    assert net._train_hip_applies(xd)
    timer = ops.KernelTimer()
    with timer:
        out = net(xd)
        (out * upd).sum().backward()
    names = timer.summary()
    assert {"conv_s2", "conv_s2_wgrad", "bn_act_stats", "bn_act_fwd", "bn_act_bwd_reduce", "bn_act_bwd_apply"} <= set(names), names
    assert ("conv_s2_dgrad" in names) == (L > 1)
    monkeypatch.setenv("EQA_CONVNET_TRAIN_MODE", "plain")
    want = ref(xd)
    (want * upd).sum().backward()
    scale = t_out.abs().max().item()
    e_hip, e_fw = (out.cpu().double() - t_out).abs().max().item() / scale, (want.cpu().double() - t_out).abs().max().item() / scale
    assert e_hip <= max(1.5 * e_fw, 2e-5), (e_hip, e_fw)
    report = {}
    for (n, p), (_, q), (_, t) in zip(net.named_parameters(), ref.named_parameters(), truth.named_parameters()):
        if n.startswith("enc_network.") and n.endswith(".bias") and int(n.split(".")[1]) % 3 == 0:
            # a conv bias behind batch statistics: the loss does not depend on it.  Exactly zero here; the fp64 evaluation holds 1e-16-size
            # noise, the framework's fp32 path up to a few % of the layer's weight-gradient scale (the rounding of a sum that cancels)
            assert p.grad.abs().max().item() == 0.0
            assert t.grad.abs().max().item() <= 1e-9 * max(dict(truth.named_parameters())[n[:-4] + "weight"].grad.abs().max().item(), 1.0)
            continue
        gs = t.grad.abs().max().item() + 1e-12
        report[n] = ((p.grad.cpu().double() - t.grad).abs().max().item() / gs, (q.grad.cpu().double() - t.grad).abs().max().item() / gs)
    bad = {n: (f"{a:.1e}", f"{b:.1e}") for n, (a, b) in report.items() if a > 1.5 * b + 2e-4}
    assert not bad, bad
    for (n, a), (_, t) in zip(net.named_buffers(), truth.named_buffers()):
        assert torch.allclose(a.cpu().double(), t.double(), atol=1e-5, rtol=1e-4), n


def test_convnetwork_training_head_dropout_drops_the_rows_the_framework_drops(dev, monkeypatch):
    """Dropout1d(0.5) on the head's 2-D input drops whole ROWS (torch reads (N, D) as an unbatched (C, L) signal); the library path
    draws the same mask from the device generator as the framework's module (same bernoulli_ / div_ on a (1, N, 1) tensor)."""
    import copy

    import equiadapt_amd as ea

    torch.manual_seed(5)
    net = ea.ConvNetwork((3, 64, 64), 16, 5, 3, 32).to(dev).train()
    ref = copy.deepcopy(net)
    x = torch.randn(64, 3, 64, 64, generator=torch.Generator().manual_seed(6)).to(dev)
    torch.cuda.manual_seed(11)
    out = net(x)
    monkeypatch.setenv("EQA_CONVNET_TRAIN_MODE", "plain")
    torch.cuda.manual_seed(11)
    want = ref(x)
    dropped = (want == ref.final_fc[3].bias).all(dim=1)            # a dropped row leaves the Linear layer's bias
    assert 8 <= int(dropped.sum()) <= 56
    assert torch.equal((out == net.final_fc[3].bias).all(dim=1), dropped)
    assert (out - want).abs().max().item() <= 2e-4 * want.abs().max().item()
    (out.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_training_kernels_take_empty_batches_and_refuse_bad_arguments(dev):
    """Edge cases of the C ABI (include/eqa_hip.h): an empty batch is a no-op (the filter gradient of nothing is zero), unsupported
    channel counts / kernel sizes are refused with EQA_ERR_UNSUPPORTED, malformed calls with EQA_ERR_INVALID_ARG -- never a launch."""
    from equiadapt_amd import _lib, ops

    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    x0 = torch.empty(0, 16, 9, 9, device=dev).permute(0, 2, 3, 1).contiguous()
    dz0 = torch.empty(0, 3, 3, 16, device=dev)
    dw = ops.conv_s2_wgrad(x0, dz0, 5, 0, False)
    assert dw.shape == (16, 16, 5, 5) and (dw == 0).all()
    assert ops.conv_s2_dgrad(dz0, ops.pack_conv_s2_dgrad_weights(torch.randn(16, 16, 5, 5, device=dev)), (9, 9), 16, 5, 0).shape == (0, 9, 9, 16)
    z0 = torch.empty(0, 16, device=dev)
    sc = torch.ones(16, device=dev)
    assert ops.bn_act_fwd(z0, sc, sc, None, 0).shape == (0, 16)
    assert not ops.conv_s2_train_supported(24, 16, 5, 0, False) and not ops.conv_s2_train_supported(16, 16, 4, 0, False)
    assert not ops.conv_s2_train_supported(5, 16, 5, 0, True) and not ops.conv_s2_train_supported(16, 16, 5, 2, False)
    assert ops.conv_s2_train_supported(3, 32, 7, 1, True) and ops.conv_s2_train_supported(64, 64, 3, 1, False)
    x = torch.randn(2, 9, 9, 16, device=dev)
    dz = torch.randn(2, 3, 3, 16, device=dev)
    out = torch.empty(16, 16, 5, 5, device=dev)
    ws = torch.empty(1 << 16, device=dev)
    assert lib.eqa_conv_s2_wgrad(x.data_ptr(), dz.data_ptr(), out.data_ptr(), ws.data_ptr(), 2, 24, 9, 9, 16, 5, 0, 0, st) == -3      # Cin % 16
    assert lib.eqa_conv_s2_wgrad(x.data_ptr(), dz.data_ptr(), out.data_ptr(), ws.data_ptr(), 2, 16, 3, 3, 16, 5, 0, 0, st) == -1      # frame < filter
    assert lib.eqa_conv_s2_wgrad(None, dz.data_ptr(), out.data_ptr(), ws.data_ptr(), 2, 16, 9, 9, 16, 5, 0, 0, st) == -1
    assert lib.eqa_conv_s2_wgrad(x.data_ptr(), dz.data_ptr(), out.data_ptr(), None, 2, 16, 9, 9, 16, 5, 0, 0, st) == -1               # no workspace
    assert lib.eqa_conv_s2_dgrad(dz.data_ptr(), ws.data_ptr(), x.data_ptr(), 2, 16, 9, 9, 16, 6, 0, st) == -3                           # K = 6
    assert lib.eqa_bn_act_fwd(x.data_ptr(), sc.data_ptr(), sc.data_ptr(), None, x.data_ptr(), 10, 6, 0, st) == -3                      # C % 4
    assert lib.eqa_bn_act_fwd(x.data_ptr(), sc.data_ptr(), sc.data_ptr(), None, x.data_ptr(), 10, 16, 2, st) == -1                     # act
    assert lib.eqa_bn_act_partial_blocks(0) == 0 and lib.eqa_bn_act_partial_blocks(2048) == 256 and lib.eqa_bn_act_partial_blocks(2048 * 900) == 2039
    assert lib.eqa_conv_s2_wgrad_workspace_bytes(0, 16, 9, 9, 16, 5, 0, 0) == 0


def test_conv_s2_gradients_fuzz(dev):
    """Sixty random shapes over every supported (Cin, Cout, K, pad, planar) family -- odd frames, frames barely larger than the
    filter, one image, runs that end inside a row -- against fp64 torch.nn.grad on the CPU."""
    import random

    from equiadapt_amd import ops

    rnd = random.Random(2025)
    fam_planar = [(c, o) for c in (1, 2, 3, 4) for o in (16, 32)]
    fam_nhwc = [(16, 16), (16, 32), (32, 32), (32, 64), (64, 64)]
    seen = set()
    for it in range(60):
        planar = rnd.random() < 0.4
        Cin, Cout = rnd.choice(fam_planar if planar else fam_nhwc)
        K, pad = rnd.choice((3, 5, 7)), rnd.choice((0, 1))
        if planar and Cin * K > 32:
            continue
        H, W = rnd.randint(K, 40), rnd.randint(K, 40)
        B = rnd.choice((1, 2, 3, 7))
        assert ops.conv_s2_train_supported(Cin, Cout, K, pad, planar)
        seen.add((planar, K, pad))
        g = torch.Generator().manual_seed(it)
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, K, K, generator=g) * 0.1
        OH, OW = (H + 2 * pad - K) // 2 + 1, (W + 2 * pad - K) // 2 + 1
        dz = torch.randn(B, Cout, OH, OW, generator=g)
        want_w = torch.nn.grad.conv2d_weight(x.double(), w.shape, dz.double(), stride=2, padding=pad)
        xd = x.to(dev) if planar else x.permute(0, 2, 3, 1).contiguous().to(dev)
        dzd = dz.permute(0, 2, 3, 1).contiguous().to(dev)
        got_w = ops.conv_s2_wgrad(xd, dzd, K, pad, planar).cpu().double()
        tol = 2e-5 * want_w.abs().max().item() + 1e-6
        assert (got_w - want_w).abs().max().item() <= tol, ("wgrad", it, planar, Cin, Cout, K, pad, H, W, B)
        if not planar:
            want_x = torch.nn.grad.conv2d_input(x.shape, w.double(), dz.double(), stride=2, padding=pad)
            got_x = ops.conv_s2_dgrad(dzd, ops.pack_conv_s2_dgrad_weights(w.to(dev)), (H, W), Cin, K, pad).permute(0, 3, 1, 2).cpu().double()
            assert (got_x - want_x).abs().max().item() <= 2e-5 * want_x.abs().max().item() + 1e-6, ("dgrad", it, Cin, Cout, K, pad, H, W, B)
            # and the forward kernel the two are the gradients of, as a vector-Jacobian identity: <conv(x), dz> == <x, dgrad(dz)>
            y = ops.conv_s2(xd, ops.pack_conv_s2_weights(w.to(dev), False), None, False, Cout, K, pad, False)
            lhs, rhs = (y.double() * dzd.double()).sum().item(), (x.double() * want_x).sum().item()
            assert abs(lhs - rhs) <= 1e-4 * (abs(rhs) + 1.0)
    assert len(seen) >= 10
