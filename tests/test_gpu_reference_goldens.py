"""Reference-GENERATED training-mode vectors through the PRODUCT on the GPU.

tests/golden/{discrete_group,vn_layers,pointcloud}.pt were produced by running the unmodified reference files
(tests/golden/make_golden.py).  tests/test_oracle_golden.py pins the CPU oracle to them; here the same vectors go through
the shipped classes on the device, so the GPU training fast paths are compared with the reference itself and not with this
repo's own op-by-op path:

  * discrete_group.pt  -> DiscreteGroupCanonicalization.groupactivations_to_groupelementonehot (train + eval, the engineered
                          tie, the straight-through gradient), prior loss, identity metric
                          (reference: equiadapt/common/basecanonicalization.py:221-256, 290-311)
  * vn_layers.pt       -> VNLinearLeakyReLU / VNBatchNorm / VNMaxPool, eval and train mode
                          (reference: pointcloud/canonicalization_networks/vector_neuron_layers.py:251-364)
  * pointcloud.pt      -> VNSmall.train() on the fused first block (csrc/vnsmall_train.hip) incl. the running statistics
                          after the step, and pooling="max" through the fused eval kernel
                          (reference: pointcloud/canonicalization_networks/equivariant_networks.py:128-150)
"""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    from equiadapt_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def test_discrete_onehot_ste_gradient_and_losses_on_product(dev, golden):
    from equiadapt_amd.common.basecanonicalization import DiscreteGroupCanonicalization

    g = golden("discrete_group.pt")
    assert g["provenance"] == "reference"

    class _Disc(DiscreteGroupCanonicalization):
        pass

    acts = g["acts"]
    d = _Disc(torch.nn.Identity(), beta=g["beta"]).to(dev)
    d.num_group = 8
    for mode in ("eval", "train"):
        d.train(mode == "train")
        a = acts.to(dev).clone().requires_grad_(True)
        oh = d.groupactivations_to_groupelementonehot(a)
        want = g["cases"][mode]["onehot"]
        if mode == "eval":
            assert torch.equal(oh.detach().cpu(), want), mode
        else:      # (hard + soft) - soft leaves a rounding residue of the softmax values (1 ulp of 1.x), device softmax != CPU's last bit
            assert torch.equal(oh.detach().cpu().argmax(-1), want.argmax(-1)) and torch.allclose(oh.detach().cpu(), want, atol=2.5e-7, rtol=0)
        if mode == "train":
            (oh * torch.arange(8.0, device=dev)).sum().backward()
            assert torch.allclose(a.grad.cpu(), g["cases"][mode]["grad"], atol=2e-7, rtol=1e-6)
        else:
            assert not oh.requires_grad          # eval mode returns the bare one-hot: no graph, as in the reference
    # the engineered tie at row 3 (columns 2 and 5): the wavefront-shuffle argmax takes the first index, like torch.argmax
    assert d.group_index(acts.to(dev))[3].item() == 2
    d.device = dev
    d.canonicalization_info_dict = {"group_activations": acts.to(dev)}
    assert torch.allclose(d.get_prior_regularization_loss().cpu(), g["prior_loss"], atol=1e-6, rtol=1e-6)
    assert torch.equal(d.get_identity_metric().cpu(), g["identity_metric"])
    d.canonicalization_info_dict["group_index"] = d.group_index(acts.to(dev))
    assert torch.equal(d.get_identity_metric().cpu(), g["identity_metric"])


def test_image_group_element_rotation_sums_match_reference_onehot(dev, golden):
    """discrete_group.py:94-135 on the product class: rotation = sum(onehot * angles), reflection = sum(onehot * indicator),
    train mode (STE one-hot) and eval mode (table lookups) against the reference-generated one-hots."""
    import equiadapt_amd as ea

    g = golden("discrete_group.pt")
    acts = g["acts"].to(dev)

    class _Net(torch.nn.Module):
        group_type, num_rotations = "roto-reflection", 4

    hp = types.SimpleNamespace(beta=g["beta"], input_crop_ratio=1.0, resize_shape=16)
    can = ea.GroupEquivariantImageCanonicalization(_Net(), hp, (3, 16, 16)).to(dev)
    ang = torch.tensor([0.0, 90.0, 180.0, 270.0] * 2)
    ind = torch.tensor([0.0] * 4 + [1.0] * 4)
    for mode in ("eval", "train"):
        can.train(mode == "train")
        el = can.groupactivations_to_groupelement(acts.clone().requires_grad_(mode == "train"))
        oh = g["cases"][mode]["onehot"]
        # train mode: the STE one-hot carries a <= 1 ulp residue of the softmax, times an angle of up to 270
        assert torch.allclose(el["rotation"].detach().cpu(), (oh * ang).sum(-1), atol=0.0 if mode == "eval" else 2e-4, rtol=0), mode
        assert torch.allclose(el["reflection"].detach().cpu(), (oh * ind).sum(-1), atol=0.0 if mode == "eval" else 1e-6, rtol=0), mode
        assert torch.equal(el["group_index"].cpu().long(), oh.argmax(-1)), mode


def test_gumbel_softmax_element_is_applied_consistently(dev):
    """gradient_trick='gumbel_softmax': the one-hot is a sample.  Image, reported rotation / reflection, masks and the invert
    must all use THAT element (the reference sums over the sampled one-hot, discrete_group.py:121-133)."""
    import equiadapt_amd as ea
    from oracle import image_ops as io

    class _FlatNet(torch.nn.Module):       # near-equal activations: the Gumbel noise decides, so sample != argmax for most images
        group_type, num_rotations = "roto-reflection", 4

        def forward(self, x):
            return x.mean(dim=(1, 2, 3))[:, None] * 1e-3 + torch.linspace(0.0, 0.01, 8, device=x.device)[None]

    hp = types.SimpleNamespace(beta=1.0, input_crop_ratio=1.0, resize_shape=16)
    can = ea.GroupEquivariantImageCanonicalization(_FlatNet(), hp, (3, 32, 32)).to(dev).eval()
    can.gradient_trick = "gumbel_softmax"      # an attribute, as in the reference (basecanonicalization.py:203-219)
    x = torch.randn(64, 3, 32, 32)
    masks = [(torch.rand(2, 32, 32) > 0.5).to(torch.uint8) for _ in range(64)]
    torch.manual_seed(5)
    with torch.no_grad():
        targets = [{"boxes": torch.tensor([[2.0, 3.0, 20.0, 25.0]], device=dev), "masks": m.to(dev)} for m in masks]
        y, t = can(x.to(dev), targets)
        info = can.canonicalization_info_dict
        gidx = info["group_index"].cpu().long()
        rot, refl = info["group_element"]["rotation"].cpu(), info["group_element"]["reflection"].cpu()
        inv = can.invert_canonicalization(torch.randn(64, 8, 32, 32, generator=torch.Generator().manual_seed(1)).to(dev))
    amax = info["group_activations"].argmax(-1).cpu()
    assert (gidx != amax).sum() >= 16, "the sample never left the argmax: the test would be vacuous"
    ang = torch.tensor([0.0, 90.0, 180.0, 270.0] * 2)
    assert torch.equal(rot, ang[gidx]) and torch.equal(refl, (gidx >= 4).float())
    assert (y.cpu() - io.canonicalize_images(x, rot, refl, (3, 32, 32))).abs().max().item() <= 1e-3
    f = torch.randn(64, 8, 32, 32, generator=torch.Generator().manual_seed(1))
    assert (inv.cpu() - io.invert_action(f, rot, refl, 4, 8, "regular")).abs().max().item() <= 1e-3
    for i in (0, 7, 33):
        want = io.rotate_masks(io.flip_masks(masks[i]), -rot[i].item())
        assert torch.equal(t[i]["masks"].cpu(), want), i
    # the identity metric stays the reference's: argmax of the activations (basecanonicalization.py:303-311)
    assert torch.equal(can.get_identity_metric().cpu(), (amax == 0).float().mean())


def test_vn_layers_on_product_match_reference(dev, golden):
    from equiadapt_amd.pointcloud.canonicalization_networks.vector_neuron_layers import VNBatchNorm, VNLinearLeakyReLU, VNMaxPool

    g = golden("vn_layers.pt")
    assert g["provenance"] == "reference"
    lay = VNLinearLeakyReLU(3, 21, dim=5, negative_slope=0.0)
    lay.load_state_dict(g["lin_state"])
    lay = lay.to(dev).eval()
    x5 = g["lin_in"].to(dev)
    with torch.no_grad():
        assert torch.allclose(lay(x5).cpu(), g["lin_out_eval"], atol=2e-6, rtol=1e-5)
    lay.train()
    out = lay(x5)                                   # grad enabled: the 3-channel maps take the fused multiply-add form
    assert torch.allclose(out.detach().cpu(), g["lin_out_train"], atol=1e-5, rtol=1e-5)
    with torch.no_grad():
        lay.load_state_dict(g["lin_state"])
        assert torch.allclose(lay(x5).cpu(), g["lin_out_train"], atol=1e-5, rtol=1e-5)       # and the GEMM form
    bn = VNBatchNorm(21, dim=4)
    bn.load_state_dict(g["bn_state"])
    bn = bn.to(dev).eval()
    with torch.no_grad():
        assert torch.allclose(bn(g["bn_in"].to(dev)).cpu(), g["bn_out_eval"], atol=2e-6, rtol=1e-5)
    mp = VNMaxPool(21)
    mp.load_state_dict(g["pool_state"])
    mp = mp.to(dev)
    with torch.no_grad():
        got = mp(g["pool_in"].to(dev)).cpu()
    # an argmax over <x, d>: a last-bit difference of the device GEMM can move a near-tie, so the picks must agree almost
    # everywhere, and wherever they differ the picked sample must still be a maximiser of the score to rounding
    same = (got == g["pool_out"]).all(dim=2)                                        # (B, C, N)
    assert same.float().mean().item() >= 0.99
    d = torch.nn.functional.linear(g["pool_in"].transpose(1, -1), g["pool_state"]["map_to_dir.weight"]).transpose(1, -1)
    score = (g["pool_in"] * d).sum(2)                                               # (B, C, N, k)
    idx_got = (g["pool_in"] == got.unsqueeze(-1)).all(dim=2).float().argmax(-1)     # which sample the device picked
    got_score = torch.gather(score, -1, idx_got.unsqueeze(-1)).squeeze(-1)
    assert (score.max(dim=-1).values - got_score).abs().max().item() <= 1e-5 * score.abs().max().item()
    from equiadapt_amd.pointcloud.canonicalization_networks.vector_neuron_layers import mean_pool
    assert torch.allclose(mean_pool(g["pool_in"].to(dev)).cpu(), g["mean_pool_out"], atol=1e-6, rtol=1e-6)


def test_vnsmall_train_mode_fused_first_block_matches_reference(dev, golden):
    """pointcloud.pt["mean_train"]: VNSmall in train() (batch statistics, dropout p = 0 as in the generator) on the fused
    first block; output and every running statistic after the step against the reference's."""
    import equiadapt_amd as ea

    t = golden("pointcloud.pt")["mean_train"]
    hp = types.SimpleNamespace(n_knn=20, pooling="mean")
    for fast in ("1", "0"):
        net = ea.VNSmall(hp)
        net.load_state_dict(t["state"])
        net.dropout.p = 0.0
        net = net.to(dev).train()
        import os
        old = os.environ.get("EQA_TRAIN_FAST")
        os.environ["EQA_TRAIN_FAST"] = fast
        try:
            out = net(t["x"].to(dev))
        finally:
            if old is None:
                os.environ.pop("EQA_TRAIN_FAST", None)
            else:
                os.environ["EQA_TRAIN_FAST"] = old
        assert out.requires_grad
        assert torch.allclose(out.detach().cpu(), t["vnsmall_out"], atol=1e-5, rtol=1e-4), fast
        after = {k: v.cpu() for k, v in net.state_dict().items()}
        for k, v in t["state_after"].items():
            if v.dtype.is_floating_point:
                assert torch.allclose(after[k], v, atol=1e-6, rtol=1e-5), (fast, k)
            else:
                assert torch.equal(after[k], v), (fast, k)
        out.sum().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_conv_network_on_product_matches_reference_golden(dev, golden):
    """Row I10 (custom_nonequivariant_networks.py:19-80): the product's ConvNetwork on the device against vectors generated by the
    reference's own class -- eval mode (the folded-batch-norm inference path), train mode with batch statistics (Dropout1d
    disabled as in the generator), the running statistics after the step and every parameter gradient."""
    import equiadapt_amd as ea

    g = golden("conv_network.pt")
    for c in g["cases"]:
        in_shape, oc, k, L, V = c["args"]
        net = ea.ConvNetwork(in_shape, oc, k, L, V)
        net.load_state_dict(c["state"])
        net = net.to(dev).eval()
        x = c["x"].to(dev)
        with torch.no_grad():
            out = net(x).cpu()
        scale = c["out_eval"].abs().max().item()
        assert (out - c["out_eval"]).abs().max().item() <= 2e-5 * scale, c["args"]
        with torch.enable_grad():                                  # eval mode with autograd: the module path
            assert (net(x).detach().cpu() - c["out_eval"]).abs().max().item() <= 2e-5 * scale, c["args"]
        net.train()
        net.final_fc[1].p = 0.0
        out = net(x)
        # train mode: batch statistics over the fixtures' 3-6 samples (BatchNorm1d) / a few positions (the last BatchNorm2d) are
        # ill-conditioned -- the device's convolution rounding is amplified up to 1e-3 here; eval mode above is the tight check
        assert (out.detach().cpu() - c["out_train"]).abs().max().item() <= 5e-3 * c["out_train"].abs().max().item(), c["args"]
        (out * c["upstream"].to(dev)).sum().backward()
        after = net.state_dict()
        for k_, v in c["state_after_train"].items():
            assert torch.allclose(after[k_].cpu().float(), v.float(), atol=1e-4, rtol=1e-3), (c["args"], k_)
        grads = {n: p.grad.cpu() for n, p in net.named_parameters()}
        if L > 3:
            # the 6-layer case ends in batch statistics over 4 x (2 x 2) positions and then over 4 samples: its gradients amplify
            # last-bit differences of the convolutions by orders of magnitude (11 % between device and CPU); outputs and running
            # statistics above are its check, the gradients only have to exist
            assert all(torch.isfinite(g_).all() for g_ in grads.values())
            continue
        for n, got in grads.items():
            want = c["grads"][n]
            if n.startswith("enc_network.") and n.endswith(".bias") and int(n.split(".")[1]) % 3 == 0:
                # a convolution bias in front of a training-mode batch-norm: the true gradient is ZERO (the sum of the batch-norm's
                # input gradient over a channel); both sides hold rounding noise of that sum, whose size follows the gradient scale of
                # the layer (B = 3 samples through BatchNorm1d batch statistics make it large), not each other: the framework's fp32
                # batch-norm backward on the device leaves 1.6e-3 of the layer's weight-gradient scale, the CPU's 5e-7
                wscale = c["grads"][n[:-4] + "weight"].abs().max().item()
                assert got.abs().max().item() <= 5e-3 * wscale + 2e-5 and want.abs().max().item() <= 5e-3 * wscale + 2e-5, (c["args"], n, wscale)
                continue
            # (the 128 x 128 case runs BatchNorm1d batch statistics over 3 samples: ill-conditioned, 6e-3 between device and CPU)
            assert (got - want).abs().max().item() <= 3e-2 * want.abs().max().item() + 2e-5, (c["args"], n)


def _knn_sets_agree(idx_dev, x, want, k):
    """Neighbour sets equal the reference's, except where the reference's own k-th / (k+1)-th candidates are closer than fp32
    can separate (evaluated in fp64): there either pick is a correct top-k."""
    got = idx_dev.long().sort(-1).values.cpu()
    ref = want.long().sort(-1).values
    bad = (got != ref).any(-1)                                   # (B, N)
    if not bad.any():
        return True, 0
    xd = x.double()
    d = (xd.transpose(1, 2)[:, :, None, :] - xd.transpose(1, 2)[:, None, :, :]).pow(2).sum(-1)     # (B, N, N)
    top = d.topk(k + 1, dim=-1, largest=False).values
    gap = (top[..., k] - top[..., k - 1]) / top[..., k].clamp_min(1e-12)
    return bool((gap[bad] < 1e-5).all()), int(bad.sum())


@pytest.mark.parametrize("grp,name", [("n1024", "mean"), ("n1024", "max"), ("k16", "mean"), ("k16", "max"), ("k8", "mean"), ("k32", "max")])
def test_vnsmall_eval_at_config4_size_and_other_k_matches_reference(dev, golden, grp, name):
    """pointcloud_n1024.pt, generated by the unmodified reference (tests/golden/make_golden_pointcloud1024.py): BASELINE config 4
    at its own size (B=8 x 1024 points, k=20) and neighbourhoods of 16 / 8 / 32 through the product in eval mode (the fused
    kernel, eqa_vnsmall_fwd) -- network output, Gram-Schmidt frame, canonical cloud, losses -- and the kNN kernel's neighbour
    sets (reference: pointcloud/canonicalization_networks/equivariant_networks.py:15-33, 128-150; continuous_group.py:51-134)."""
    import equiadapt_amd as ea
    from equiadapt_amd import _lib

    c = golden("pointcloud_n1024.pt")[grp][name]
    k, B, N = c["k"], c["B"], c["N"]
    hp = types.SimpleNamespace(n_knn=k, pooling=c["pooling"])
    net = ea.VNSmall(hp)
    net.load_state_dict(c["state"])
    net = net.to(dev).eval()
    x = c["x"].to(dev)
    lib = _lib.load()
    idx = torch.empty(B, N, k, dtype=torch.int32, device=dev)
    _lib.check(lib.eqa_vn_knn(x.data_ptr(), idx.data_ptr(), B, N, k, None), "eqa_vn_knn")
    torch.cuda.synchronize()
    ok, n_bad = _knn_sets_agree(idx, c["x"], c["knn_idx"], k)
    assert ok, f"{n_bad} points disagree outside fp32 near-ties"
    # best first, like torch.topk: the reference's own ORDER wherever its consecutive scores are distinct in fp32
    same_order = (idx.long().cpu() == c["knn_idx"].long()).float().mean().item()
    assert same_order >= 0.999, same_order
    with torch.no_grad():
        vec = net(x).cpu()
    # max pooling: one argmax pick moved by a last-bit score difference changes the mean over N points by |dx| / N
    tol = (2e-6 if c["pooling"] == "mean" else 2e-5) * max(c["vnsmall_out"].abs().max().item(), 1.0)
    assert (vec - c["vnsmall_out"]).abs().max().item() <= tol + 1e-4 * tol, (grp, name, (vec - c["vnsmall_out"]).abs().max().item())
    can = ea.EquivariantPointcloudCanonicalization(net, hp).to(dev).eval()
    with torch.no_grad():
        xc = can(x).cpu()
    R = can.canonicalization_info_dict["group_element_matrix_representation"].cpu()
    if c["pooling"] == "mean":
        assert (R - c["rotation"]).abs().max().item() <= 1e-4
        assert (xc - c["x_canonicalized"]).abs().max().item() <= 5e-4
    else:
        # max pooling: the network output may differ by moved argmax picks (bound above); Gram-Schmidt amplifies that by the
        # conditioning of the three vectors (measured 3.8e-4 on one of the eight clouds).  So: the frame is exactly the
        # Gram-Schmidt of the output the product computed (the kernel's own arithmetic, 2e-5), the canonical cloud is that frame
        # applied to the cloud, and both stay within the amplified bound of the reference's
        from oracle import pointcloud_ops as po

        assert (R - po.gram_schmidt(vec)).abs().max().item() <= 2e-5
        assert (xc - po.canonicalize_pointcloud(c["x"], R)).abs().max().item() <= 1e-5
        assert (R - c["rotation"]).abs().max().item() <= 2e-3
        assert (xc - c["x_canonicalized"]).abs().max().item() <= 1e-2
    ltol = 1e-4 if c["pooling"] == "mean" else 2e-3
    assert torch.allclose(can.get_prior_regularization_loss().cpu(), c["prior_loss"], atol=ltol)
    assert torch.allclose(can.get_identity_metric().cpu(), c["identity_metric"], atol=ltol)


@pytest.mark.parametrize("grp", ["n1024", "k16"])
def test_vnsmall_training_step_at_config4_size_matches_reference(dev, golden, grp):
    """The training-mode forward + backward (fused first block and tail, csrc/vnsmall_train.hip / vnsmall_tail.hip) at B=8 x 1024
    points against the reference's own step: output, running statistics after the step, parameter gradients of sum(out * w).
    A VN-ReLU gate (<q, d> >= 0) that flips on a last-bit difference moves a gradient entry by O(1/(B N)) of its scale; with
    8192 points the bound is 1 % of each gradient's scale."""
    import equiadapt_amd as ea

    t = golden("pointcloud_n1024.pt")[grp]["mean_train"]
    hp = types.SimpleNamespace(n_knn=t["k"], pooling="mean")
    net = ea.VNSmall(hp)
    net.load_state_dict(t["state"])
    net.dropout.p = 0.0
    net = net.to(dev).train()
    out = net(t["x"].to(dev))
    # the output is a mean over the points of O(1) vectors that largely cancel (|out| ~ 0.03): the error scale is the summands',
    # and one VN-ReLU gate flipped by a last-bit difference moves the mean by O(1 / N)
    err = (out.detach().cpu() - t["vnsmall_out"]).abs().max().item()
    assert err <= 5e-5, err
    (out * t["w"].to(dev)).sum().backward()
    after = {k: v.cpu() for k, v in net.state_dict().items()}
    for k, v in t["state_after"].items():
        if v.dtype.is_floating_point:
            assert torch.allclose(after[k], v, atol=1e-6, rtol=1e-5), k
        else:
            assert torch.equal(after[k], v), k
    # Most of these gradients are fp32-noise dominated: the loss is a mean of vectors that cancel, and a gradient along a weight
    # scale in front of a batch-norm is zero in exact arithmetic -- the reference's own fp32 values are 2.5 % (8 x 1024 points) to
    # 9 % (4 x 512) of the tensor's scale away from an fp64 evaluation of the same step (tools/diag/vn_train_grads.py), the
    # framework's op-by-op fp32 path on this device up to 22 %.  So the truth is the fp64 op-by-op evaluation, and the product must
    # be as close to it as the reference is (x4, floor 2e-3); the fused first block's backward itself is checked tightly, with a
    # well-conditioned upstream gradient, in test_gpu_backward.py::test_vnsmall_training_fast_path_matches_op_path.
    import os

    net64 = ea.VNSmall(hp)
    net64.load_state_dict(t["state"])
    net64.dropout.p = 0.0
    net64 = net64.to(dev).double().train()
    old = os.environ.get("EQA_TRAIN_FAST")
    os.environ["EQA_TRAIN_FAST"] = "0"
    try:
        (net64(t["x"].to(dev).double()) * t["w"].to(dev).double()).sum().backward()
    finally:
        if old is None:
            os.environ.pop("EQA_TRAIN_FAST", None)
        else:
            os.environ["EQA_TRAIN_FAST"] = old
    truth = {n: p.grad.cpu() for n, p in net64.named_parameters() if p.grad is not None}
    for n, p in net.named_parameters():
        want = t["grads"].get(n)
        if want is None:
            assert p.grad is None or p.grad.abs().max().item() == 0, n
            continue
        scale = max(truth[n].abs().max().item(), 1e-6)
        ref_err = (want.double() - truth[n]).abs().max().item() / scale
        got_err = (p.grad.cpu().double() - truth[n]).abs().max().item() / scale
        assert got_err <= max(4.0 * ref_err, 2e-3), (n, got_err, ref_err)
