"""eqa_lift5_fft48k5_input (round 6, csrc/lift_fft.hip): the lifting convolution fused into the forward FFT-48 transform of the
layer behind it, against (1) the two kernels it replaces -- eqa_lift_conv_grouped then eqa_fft48k5_input_grouped -- on the same
operands, (2) an fp64 evaluation of relu(conv2d(x, bank) + bias) carried through torch.fft, and (3) the whole two-layer
convolution against F.conv2d in fp64.  Reference layers: escnn_networks.py:60-85."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _spectra_fp64(y: torch.Tensor):
    """(nimg, C, H1, W1) fp64 map -> the (F, M, C) complex spectra of its 48 x 48 tiles at stride 44 (zero beyond the map)."""
    from equiadapt_amd.images.canonicalization_networks import fftconv

    nimg, C, H1, W1 = y.shape
    TY, TX = fftconv.tiles(H1), fftconv.tiles(W1)
    yp = torch.zeros(nimg, C, 44 * (TY - 1) + 48, 44 * (TX - 1) + 48, dtype=y.dtype, device=y.device)
    yp[:, :, :H1, :W1] = y
    tiles = torch.stack([yp[:, :, 44 * ty:44 * ty + 48, 44 * tx:44 * tx + 48] for ty in range(TY) for tx in range(TX)], dim=1)   # (nimg, T, C, 48, 48)
    spec = torch.fft.rfft2(tiles)                                                                     # (nimg, T, C, 48, 25)
    ky, kx = fftconv.freq_index()
    return spec[..., ky.to(y.device), kx.to(y.device)].reshape(nimg * TY * TX, C, -1).permute(2, 0, 1)  # (F, M, C)


def _unpack_V(V: torch.Tensor, C: int):
    """(F, M, 2C) fp32 rows in [Re x 16 | Im x 16] groups -> (F, M, C) complex128."""
    Fq, M, _ = V.shape
    v = V.double().reshape(Fq, M, C // 16, 2, 16)
    return torch.complex(v[:, :, :, 0], v[:, :, :, 1]).reshape(Fq, M, C)


CASES = [(3, 96, 96, 64), (2, 60, 75, 32), (1, 52, 52, 16), (2, 100, 97, 48), (1, 140, 53, 32), (5, 96, 96, 256)]


@pytest.mark.parametrize("nimg,H0,W0,C", CASES)
@pytest.mark.parametrize("relu,with_bias", [(True, True), (False, False)])
def test_fused_lift_fft_input_matches_the_two_kernels_and_fp64(dev, nimg, H0, W0, C, relu, with_bias):
    from equiadapt_amd import _lib, ops
    from equiadapt_amd.images.canonicalization_networks import fftconv

    lib = _lib.load()
    assert lib.eqa_lift5_fft48k5_input_supported(3, 5, 5, C) and not lib.eqa_lift5_fft48k5_input_supported(3, 5, 5, 24) \
        and not lib.eqa_lift5_fft48k5_input_supported(1, 5, 5, 64) and not lib.eqa_lift5_fft48k5_input_supported(3, 3, 3, 64)
    g = torch.Generator().manual_seed(nimg * 1000 + H0 + C)
    x = torch.randn(nimg, 3, H0, W0, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    bank = (torch.randn(C, 3, 5, 5, generator=g) / 75 ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(C, generator=g).to(dev) if with_bias else None
    H1, W1 = H0 - 4, W0 - 4
    M = nimg * fftconv.tiles(H1) * fftconv.tiles(W1)
    pitch = lib.eqa_fft48k5_tile_pitch(M)
    st = torch.cuda.current_stream().cuda_stream
    # fused, into a poisoned buffer: the padding row of every frequency stays untouched
    full = torch.full((fftconv.F, pitch, 2 * C), 7.0, dtype=torch.float32, device=dev)
    _lib.check(lib.eqa_lift5_fft48k5_input(x.data_ptr(), bank.data_ptr(), bias.data_ptr() if with_bias else None, int(relu), full.data_ptr(),
                                           nimg, H0, W0, C, st), "eqa_lift5_fft48k5_input")
    assert (full[:, M:] == 7.0).all()
    got = full[:, :M]
    # (2) fp64 truth
    y64 = F.conv2d(x.double(), bank.double(), bias.double() if with_bias else None)
    y64 = torch.relu(y64) if relu else y64
    want = _spectra_fp64(y64)
    scale = want.abs().max().item()
    err = (_unpack_V(got, C) - want).abs().max().item()
    assert err <= 3e-6 * scale, (err, scale)
    # (3) the same on the bf16 matrix cores (exact three-piece splits, six products): as close to fp64 as the fp32 matrix instruction
    wp = fftconv.LiftedInput(x, bank, bias, relu).pieces()
    assert wp.shape == (C, 3, 16, 8) and wp.dtype == torch.bfloat16 and wp.numel() * 2 == lib.eqa_lift5_pieces_bytes(C)
    full_p = torch.full((fftconv.F, pitch, 2 * C), 7.0, dtype=torch.float32, device=dev)
    _lib.check(lib.eqa_lift5_fft48k5_input_bf16x3(x.data_ptr(), wp.data_ptr(), bias.data_ptr() if with_bias else None, int(relu), full_p.data_ptr(),
                                                  nimg, H0, W0, C, st), "eqa_lift5_fft48k5_input_bf16x3")
    assert (full_p[:, M:] == 7.0).all()
    err_p = (_unpack_V(full_p[:, :M], C) - want).abs().max().item()
    print(f"fused lift+fft {(nimg, H0, W0, C)} relu={relu}: |bf16x3 - fp64| {err_p:.3e}, |f32 - fp64| {err:.3e}, scale {scale:.3e}")
    assert err_p <= 3e-6 * scale and err_p <= 1.5 * err + 1e-7 * scale, (err_p, err, scale)
    # (1) the two kernels it replaces (where the unfused lifting kernel takes the channel count)
    if C % 64 == 0 and ops.lift_conv_supported(3, 5, 5, C):
        ymap = ops.lift_conv_grouped(x, ops.pack_lift_weights(bank), bias, relu, 5, 5)
        Vref = fftconv.spectra_buffer(M, 2 * C, dev)
        T = torch.empty(max(lib.eqa_fft48k5_workspace_bytes(nimg, H1, W1 - 4, C), 4) // 4, dtype=torch.float32, device=dev)
        _lib.check(lib.eqa_fft48k5_input_grouped(ymap.data_ptr(), T.data_ptr(), Vref.data_ptr(), None, 0, nimg, H1, W1, C, st), "input_grouped")
        err_ref = (_unpack_V(Vref, C) - want).abs().max().item()
        d = (got - Vref).abs().max().item()
        print(f"fused lift+fft {(nimg, H0, W0, C)} relu={relu}: |fused - fp64| {err:.3e}, |two kernels - fp64| {err_ref:.3e}, |fused - two kernels| {d:.3e}, "
              f"scale {scale:.3e}, bit-equal {torch.equal(got, Vref)}")
        assert d <= 2e-6 * scale and err <= 1.5 * err_ref + 1e-7 * scale


def test_conv5x5_takes_a_lifted_input(dev, monkeypatch):
    """fftconv.conv5x5 on a LiftedInput (the fused kernel as its input transform) = conv2d(relu(conv2d(x, w1) + b1), w2) + b2 in fp64,
    both as the map and as the window sums in front of a linearised tail; and the canonicalization network's inference path takes
    the fused route by default and equals its own unfused route (EQA_LIFT_FFT_FUSED=0 semantics: fftconv.lift_fused_applicable)."""
    from equiadapt_amd.images.canonicalization_networks import fftconv

    torch.manual_seed(61)
    monkeypatch.setenv("EQA_LIFT_FFT_FUSED", "1")
    for (nimg, H0, W0, C1, C2) in [(9, 96, 96, 64, 64), (6, 96, 140, 32, 128)]:
        x = torch.randn(nimg, 3, H0, W0, device=dev).contiguous(memory_format=torch.channels_last)
        w1 = (torch.randn(C1, 3, 5, 5, device=dev) / 75 ** 0.5).contiguous(memory_format=torch.channels_last)
        b1 = torch.randn(C1, device=dev)
        w2 = torch.randn(C2, C1, 5, 5, device=dev) / (5 * C1 ** 0.5)
        b2 = torch.randn(C2, device=dev)
        assert fftconv.lift_fused_applicable(x.shape, w1.shape, C2, dev)
        h = fftconv.LiftedInput(x, w1, b1, True)
        assert h.shape == (nimg, C1, H0 - 4, W0 - 4)
        Bm = fftconv.spectra_for(w2)
        got = fftconv.conv5x5(h, Bm, b2, True, None, False)
        want = torch.relu(F.conv2d(torch.relu(F.conv2d(x.double(), w1.double(), b1.double())), w2.double(), b2.double()))
        scale = want.abs().max().item()
        assert got.shape == want.shape and (got.double() - want).abs().max().item() <= 5e-6 * scale
        mat = h.materialize()
        assert (mat.double() - torch.relu(F.conv2d(x.double(), w1.double(), b1.double()))).abs().max().item() <= 2e-6 * mat.abs().max().item()
        OH, OW = H0 - 8, W0 - 8
        S = fftconv.conv5x5(h, Bm, b2, True, None, False, sums_k=5)
        Sw = torch.stack([torch.stack([want[:, :, u:u + OH - 4, v:v + OW - 4].sum((-1, -2)) for v in range(5)], -1) for u in range(5)], -2)
        assert ((S - Sw).abs().max() <= 2e-6 * Sw.abs().max().clamp_min(scale))


def test_network_inference_takes_the_fused_route(dev, monkeypatch):
    import equiadapt_amd as ea
    from equiadapt_amd.images.canonicalization_networks import fftconv

    torch.manual_seed(5)
    net = ea.ESCNNEquivariantNetwork((3, 96, 96), 8, 5, "rotation", 8, 3).to(dev).eval()     # 8 fields x 8 = 64 channels
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(12, 3, 96, 96, device=dev)
    calls = []
    real = fftconv.LiftedInput.__init__

    def spy(self, *a, **k):
        calls.append(1)
        return real(self, *a, **k)

    monkeypatch.setattr(fftconv.LiftedInput, "__init__", spy)
    monkeypatch.setenv("EQA_LIFT_FFT_FUSED", "1")
    with torch.no_grad():
        fused = net(x)
    assert calls, "the inference path did not build a LiftedInput"
    monkeypatch.setenv("EQA_LIFT_FFT_FUSED", "0")
    with torch.no_grad():
        unfused = net(x)
    assert len(calls) == 1
    assert (fused - unfused).abs().max().item() <= 2e-6 * unfused.abs().max().item()
    assert torch.equal(fused.argmax(1), unfused.argmax(1))
    # the default form of the fused kernel is the fp16 one, its operand cached per weight version; the fp32 form gives the same activations
    assert fftconv.LIFT_FFT_FORM == "h2" and ("lifth", id(net.eqv_network[0])) in net._fold_cache
    monkeypatch.setenv("EQA_LIFT_FFT_FUSED", "1")
    monkeypatch.setattr(fftconv, "LIFT_FFT_FORM", "f32")
    with torch.no_grad():
        f32_form = net(x)
    assert (f32_form - unfused).abs().max().item() <= 2e-6 * unfused.abs().max().item() and torch.equal(f32_form.argmax(1), unfused.argmax(1))
    # the opt-in form with the convolution on the bf16 matrix cores (exact three-piece splits): same activations, pieces cached per weights
    monkeypatch.setenv("EQA_LIFT_FFT_FUSED", "1")
    monkeypatch.setattr(fftconv, "LIFT_FFT_FORM", "bf16x3")
    with torch.no_grad():
        pieces_form = net(x)
        again = net(x)
    assert ("liftp", id(net.eqv_network[0])) in net._fold_cache and torch.equal(pieces_form, again)
    assert (pieces_form - unfused).abs().max().item() <= 2e-6 * unfused.abs().max().item()
    assert torch.equal(pieces_form.argmax(1), unfused.argmax(1))


@pytest.mark.parametrize("nimg,H0,W0,C", [(3, 96, 96, 64), (2, 100, 97, 48), (1, 52, 52, 16), (5, 96, 96, 256), (40, 96, 96, 256)])
def test_fused_kernel_hands_over_a_bound_of_its_spectra(dev, nimg, H0, W0, C):
    """eqa_lift5_fft48k5_input_dcmax: the same spectra as eqa_lift5_fft48k5_input, bit for bit, plus EQA_LIFT5_DCMAX_SLOTS floats whose
    maximum is the largest DC bin stored -- which bounds every |Re|, |Im| of V when the activations are non-negative (|X[k]| <= X[0]):
    the `vbound` of the fp16 contraction.  Slots of blocks that do not exist are 0; relu = 0 is refused."""
    from equiadapt_amd import _lib
    from equiadapt_amd.images.canonicalization_networks import fftconv

    lib = _lib.load()
    g = torch.Generator().manual_seed(nimg + H0 + C)
    x = torch.randn(nimg, 3, H0, W0, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    bank = (torch.randn(C, 3, 5, 5, generator=g) / 75 ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(C, generator=g).to(dev)
    M = nimg * fftconv.tiles(H0 - 4) * fftconv.tiles(W0 - 4)
    pitch = lib.eqa_fft48k5_tile_pitch(M)
    st = torch.cuda.current_stream().cuda_stream
    plain = torch.full((fftconv.F, pitch, 2 * C), 7.0, dtype=torch.float32, device=dev)
    _lib.check(lib.eqa_lift5_fft48k5_input(x.data_ptr(), bank.data_ptr(), bias.data_ptr(), 1, plain.data_ptr(), nimg, H0, W0, C, st), "plain")
    full = torch.full((fftconv.F, pitch, 2 * C), 7.0, dtype=torch.float32, device=dev)
    slots = torch.full((fftconv.DCMAX_SLOTS + 8,), -3.0, dtype=torch.float32, device=dev)
    _lib.check(lib.eqa_lift5_fft48k5_input_dcmax(x.data_ptr(), bank.data_ptr(), bias.data_ptr(), 1, full.data_ptr(), slots.data_ptr(), nimg, H0, W0,
                                                 C, st), "dcmax")
    assert torch.equal(full, plain)
    assert (slots[fftconv.DCMAX_SLOTS:] == -3.0).all() and (slots[:fftconv.DCMAX_SLOTS] >= 0).all()
    V = full[:, :M]
    dc = V[48 * 23]                                      # frequency (kx = 0, ky = 0): [Re x 16 | Im x 16] per channel group
    assert slots[:fftconv.DCMAX_SLOTS].max().item() == dc.max().item()
    assert V.abs().max().item() <= slots[:fftconv.DCMAX_SLOTS].max().item()
    assert lib.eqa_lift5_fft48k5_input_dcmax(x.data_ptr(), bank.data_ptr(), bias.data_ptr(), 0, full.data_ptr(), slots.data_ptr(), nimg, H0, W0, C,
                                             st) == -3        # EQA_ERR_UNSUPPORTED


def test_two_layer_convolution_takes_the_fp16_contraction_behind_the_fused_kernel(dev):
    """conv5x5 on a LiftedInput with relu, Cin = 64 / 256, Cout = 128 / 256: the fused kernel hands its DC bins to the fp16 form of
    the contraction ("h3"); the result against F.conv2d in fp64 at the tolerance of the fp32 path, and no further from it than the
    same chain with the fp32 matrix instruction."""
    from equiadapt_amd.images.canonicalization_networks import fftconv

    for (nimg, C1, C2, scale_in) in [(9, 64, 128, 1.0), (9, 256, 256, 1.0), (9, 256, 256, 3000.0), (9, 64, 128, 1e-3)]:
        g = torch.Generator().manual_seed(C1 + C2)
        x = (torch.randn(nimg, 3, 96, 96, generator=g) * scale_in).to(dev).contiguous(memory_format=torch.channels_last)
        bank1 = (torch.randn(C1, 3, 5, 5, generator=g) / 75 ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
        b1 = (torch.randn(C1, generator=g) * scale_in).to(dev)
        w2 = (torch.randn(C2, C1, 5, 5, generator=g) / (5 * C1 ** 0.5)).to(dev)
        b2 = torch.randn(C2, generator=g).to(dev)
        B2 = fftconv.spectra_for(w2)
        want = F.conv2d(torch.relu(F.conv2d(x.double(), bank1.double(), b1.double())), w2.double(), b2.double())
        lifted = fftconv.LiftedInput(x, bank1, b1, True)
        got = fftconv.conv5x5(lifted, B2, b2, False)
        assert fftconv.LAST_FORM == "h3"
        keep = fftconv.GEMM_PIECES
        try:
            fftconv.GEMM_PIECES = "f32"
            ref = fftconv.conv5x5(lifted, B2, b2, False)
            assert fftconv.LAST_FORM == "f32"
        finally:
            fftconv.GEMM_PIECES = keep
        s = want.abs().max().item()
        e_h3, e_f32 = (got.double() - want).abs().max().item(), (ref.double() - want).abs().max().item()
        print(f"two-layer chain C {C1} -> {C2}, input scale {scale_in:g}: |h3 - fp64| {e_h3 / s:.3e}, |f32 - fp64| {e_f32 / s:.3e} (of max |y|)")
        assert e_h3 <= 5e-6 * s and e_h3 <= 1.25 * e_f32 + 1e-7 * s, (C1, C2, scale_in, e_h3 / s, e_f32 / s)


@pytest.mark.parametrize("nimg,H0,W0,C", CASES + [(40, 96, 96, 256)])
@pytest.mark.parametrize("relu,with_bias,xscale", [(True, True, 1.0), (False, False, 1.0), (True, True, 300.0), (True, False, 1e-3)])
def test_fused_kernel_on_two_fp16_pieces_matches_fp64_like_the_fp32_form(dev, nimg, H0, W0, C, relu, with_bias, xscale):
    """eqa_lift5_fft48k5_input_f16x2 (FORM 2 of csrc/lift_fft.hip: pixels and weights as two fp16 pieces, three exact products,
    scaled by powers of two under eqa_absmax_slots' bound) against an fp64 evaluation at the fp32 form's tolerance, no further from
    it than the fp32 form (x 1.5: both sit at a few fp32 ulps of the largest bin), the padding rows untouched, the DC slots equal to the
    stored DC bins, at input scales 1e-3 .. 300."""
    from equiadapt_amd import _lib
    from equiadapt_amd.images.canonicalization_networks import fftconv

    lib = _lib.load()
    g = torch.Generator().manual_seed(nimg * 1000 + H0 + C)
    x = (torch.randn(nimg, 3, H0, W0, generator=g) * xscale).to(dev).contiguous(memory_format=torch.channels_last)
    bank = (torch.randn(C, 3, 5, 5, generator=g) / 75 ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    bias = (torch.randn(C, generator=g) * xscale).to(dev) if with_bias else None
    p_b = bias.data_ptr() if with_bias else None
    M = nimg * fftconv.tiles(H0 - 4) * fftconv.tiles(W0 - 4)
    pitch = lib.eqa_fft48k5_tile_pitch(M)
    st = torch.cuda.current_stream().cuda_stream
    lifted = fftconv.LiftedInput(x, bank, bias, relu)
    wh, w_scale = lifted.pieces_f16()
    assert wh.shape == (C, 2, 5, 4, 8) and wh.dtype == torch.float16 and wh.numel() * 2 == lib.eqa_lift5_pieces_f16_bytes(C)
    assert (wh[:, :, :, 3] == 0).all() and (wh[..., 3] == 0).all() and (wh[..., 7] == 0).all() and (wh[:, :, :, 0, :4] == 0).all()
    xb = torch.full((fftconv.DCMAX_SLOTS,), -1.0, dtype=torch.float32, device=dev)
    _lib.check(lib.eqa_absmax_slots(x.data_ptr(), x.numel(), xb.data_ptr(), st), "absmax")
    assert xb.max().item() == x.abs().max().item() and (xb >= 0).all()
    full = torch.full((fftconv.F, pitch, 2 * C), 7.0, dtype=torch.float32, device=dev)
    slots = torch.full((fftconv.DCMAX_SLOTS,), -3.0, dtype=torch.float32, device=dev)
    _lib.check(lib.eqa_lift5_fft48k5_input_f16x2(x.data_ptr(), wh.data_ptr(), w_scale, xb.data_ptr(), fftconv.DCMAX_SLOTS, p_b, int(relu),
                                                 full.data_ptr(), slots.data_ptr() if relu else None, nimg, H0, W0, C, st), "f16x2")
    assert (full[:, M:] == 7.0).all()
    ref = torch.full((fftconv.F, pitch, 2 * C), 7.0, dtype=torch.float32, device=dev)
    _lib.check(lib.eqa_lift5_fft48k5_input(x.data_ptr(), bank.data_ptr(), p_b, int(relu), ref.data_ptr(), nimg, H0, W0, C, st), "f32")
    y64 = F.conv2d(x.double(), bank.double(), bias.double() if with_bias else None)
    y64 = torch.relu(y64) if relu else y64
    want = _spectra_fp64(y64)
    scale = want.abs().max().item()
    e_h = (_unpack_V(full[:, :M], C) - want).abs().max().item()
    e_f = (_unpack_V(ref[:, :M], C) - want).abs().max().item()
    print(f"fused lift+fft f16x2 {(nimg, H0, W0, C)} relu={relu} x{xscale:g}: |f16x2 - fp64| {e_h / scale:.3e}, |f32 - fp64| {e_f / scale:.3e} (of max |V|)")
    assert e_h <= 3e-6 * scale and e_h <= 1.5 * e_f + 1e-7 * scale, (e_h / scale, e_f / scale)
    if relu:
        assert slots.max().item() == full[48 * 23, :M].max().item() and full[:, :M].abs().max().item() <= slots.max().item()
        assert lib.eqa_lift5_fft48k5_input_f16x2(x.data_ptr(), wh.data_ptr(), w_scale, xb.data_ptr(), 256, p_b, 0, full.data_ptr(), slots.data_ptr(),
                                                 nimg, H0, W0, C, st) == -3
    assert lib.eqa_lift5_fft48k5_input_f16x2(x.data_ptr(), wh.data_ptr(), 3.0, xb.data_ptr(), 256, p_b, int(relu), full.data_ptr(), None,
                                             nimg, H0, W0, C, st) == -1        # the weights' scale must be a power of two
