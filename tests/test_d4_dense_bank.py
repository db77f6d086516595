"""The dense-export bridge (``ESCNNEquivariantNetwork.load_exported_dense``, the route for e2cnn-trained weights; reference network
escnn_networks.py:48-91) for configs[4]'s group, D4, with a bank built here from the definition of the group convolution -- the
counterpart of test_gpu_parity.py::test_load_exported_dense_with_an_independently_built_regular_bank (C4)."""
import types

import pytest
import torch


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    from equiadapt_amd import _lib

    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def _d4_tables():
    """D4 as spatial operators on square arrays, element index e = 4 m + r  <->  T_e = hflip^m o rot90^r (quarter turns first, then
    the flip): the reading under which the reference's canonicalize, x -> hflip^m -> rotate(-90 r)
    (images/canonicalization/discrete_group.py:209-213 with rotation = cat([ang, ang])[e], reflection = [0,0,0,0,1,1,1,1][e], :127-133),
    is T_e^-1.  Multiplication and inverses are found NUMERICALLY by applying the operators to a generic array."""
    def T(e, a):
        m, r = divmod(e, 4)
        a = torch.rot90(a, r, (-2, -1))
        return a.flip(-1) if m else a

    probe = torch.arange(25.0).reshape(5, 5)
    images = [T(e, probe) for e in range(8)]
    mul = torch.empty(8, 8, dtype=torch.long)                      # T_a o T_b = T_mul[a, b]
    for a in range(8):
        for b in range(8):
            mul[a, b] = next(c for c in range(8) if torch.equal(T(a, T(b, probe)), images[c]))
    inv = torch.tensor([next(b for b in range(8) if mul[a, b] == 0) for a in range(8)])
    return T, mul, inv


def _d4_regular_dense_layers(fields: int, k: int, layers: int, gen: torch.Generator):
    """A D4 regular-representation G-CNN in the exported dense form (channel = field * 8 + element), from the definition of the
    group convolution -- W[(o, h), (i, h')] = T_h( w[o, i, h^-1 h'] ), lifting W[(o, h), c] = T_h( w[o, c] ) -- with exact quarter
    turns and flips; nothing of the product's filter-bank code is used."""
    import torch.nn as nn

    T, mul, inv = _d4_tables()
    convs, norms = [], []
    cin = 3
    for layer in range(layers):
        if layer == 0:
            w = torch.randn(fields, 3, k, k, generator=gen) * (1.0 / (3 * k * k)) ** 0.5
            W = torch.stack([T(h, w) for h in range(8)], dim=1).reshape(fields * 8, 3, k, k)
        else:
            w = torch.randn(fields, fields, 8, k, k, generator=gen) * (1.0 / (fields * 8 * k * k)) ** 0.5
            W = torch.empty(fields, 8, fields, 8, k, k)
            for h in range(8):
                for h2 in range(8):
                    W[:, h, :, h2] = T(h, w[:, :, int(mul[inv[h], h2])])
            W = W.reshape(fields * 8, fields * 8, k, k)
        cv = nn.Conv2d(cin, fields * 8, k, bias=True)
        with torch.no_grad():
            cv.weight.copy_(W)
            cv.bias.copy_((torch.randn(fields, generator=gen) * 0.1).repeat_interleave(8))
        convs.append(cv)
        cin = fields * 8
        if layer < layers - 1:
            bn = nn.BatchNorm2d(fields * 8)
            with torch.no_grad():
                bn.weight.copy_((torch.rand(fields, generator=gen) + 0.5).repeat_interleave(8))
                bn.bias.copy_((torch.randn(fields, generator=gen) * 0.2).repeat_interleave(8))
                bn.running_mean.copy_((torch.randn(fields, generator=gen) * 0.2).repeat_interleave(8))
                bn.running_var.copy_((torch.rand(fields, generator=gen) + 0.5).repeat_interleave(8))
            norms.append(bn.eval())
    return convs, norms


def _dense_activations(convs, norms, fields, G, x64):
    h = x64
    for i, cv in enumerate(convs):
        h = torch.nn.functional.conv2d(h, cv.weight.double(), cv.bias.double())
        if i < len(norms):
            bn = norms[i]
            h = (h - bn.running_mean.double()[None, :, None, None]) / torch.sqrt(bn.running_var.double()[None, :, None, None] + bn.eps)
            h = torch.relu(h * bn.weight.double()[None, :, None, None] + bn.bias.double()[None, :, None, None])
    return h.reshape(h.shape[0], fields, G, h.shape[-2], h.shape[-1]).mean(dim=(1, 3, 4))


def test_d4_tables_and_dense_bank_are_a_regular_representation():
    """CPU: the test's own D4 construction before it is trusted on the GPU -- the numerically found table is a group (closed,
    associative, inverses), and transforming the input by g permutes the dense network's pooled activations as the regular
    representation does, a'[h] = a[g^-1 h], for every g (fp64)."""
    T, mul, inv = _d4_tables()
    for a in range(8):
        assert sorted(mul[a].tolist()) == list(range(8)) and mul[a, inv[a]] == 0 and mul[inv[a], a] == 0
        for b in range(8):
            for c in range(8):
                assert mul[mul[a, b], c] == mul[a, mul[b, c]]
    assert mul[4, 1] != mul[1, 4]                                   # not abelian: the order "turn, then flip" matters
    gen = torch.Generator().manual_seed(99)
    convs, norms = _d4_regular_dense_layers(4, 5, 3, gen)
    x = torch.randn(3, 3, 40, 40, generator=gen).double()
    with torch.no_grad():
        a0 = _dense_activations(convs, norms, 4, 8, x)
        for g in range(8):
            ag = _dense_activations(convs, norms, 4, 8, T(g, x))
            want = a0[:, [int(mul[inv[g], h]) for h in range(8)]]
            assert (ag - want).abs().max().item() <= 1e-12 * a0.abs().max().item() + 1e-13, g


@pytest.mark.gpu
@pytest.mark.parametrize("fields,size", [(8, 96), (4, 60)])
def test_load_exported_dense_with_an_independently_built_d4_bank(dev, fields, size):
    """configs[4]'s group (D4 = roto-reflection, num_rotations 4) through the bridge for e2cnn-trained weights (reference network:
    escnn_networks.py:48-91) with weights the product did NOT export: the dense D4 regular-representation network above.
    (1) [CPU test above] it is what it claims to be; (2) the product loaded with it reproduces the fp64 evaluation through the
    inference fast path and the module path; (3) a canonicalizer on it picks the dense network's own argmax; (4) the element
    convention is the reference's: the canonical image of T_g x equals that of x for a quarter turn and for the flip."""
    import copy

    import equiadapt_amd as ea

    T, mul, inv = _d4_tables()
    gen = torch.Generator().manual_seed(4321 + fields)
    k, layers, G = 5, 3, 8
    convs, norms = _d4_regular_dense_layers(fields, k, layers, gen)
    x = torch.randn(10, 3, size, size, generator=gen)
    with torch.no_grad():
        want = _dense_activations(convs, norms, fields, G, x.double())
    net = ea.ESCNNEquivariantNetwork((3, size, size), fields, k, "roto-reflection", 4, layers).to(dev).eval()
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
        y0 = can(x.to(dev)).cpu()
    g0 = can.canonicalization_info_dict["group_index"].cpu().long()
    assert torch.equal(g0[clear], want.argmax(-1)[clear])
    el = can.canonicalization_info_dict["group_element"]
    assert torch.equal(el["rotation"].cpu(), (g0 % 4).float() * 90.0) and torch.equal(el["reflection"].cpu(), (g0 >= 4).float())
    for g in (1, 4, 6):                                             # a quarter turn, the flip, a flip + half turn
        with torch.no_grad():
            yg = can(T(g, x).contiguous().to(dev)).cpu()
        gg = can.canonicalization_info_dict["group_index"].cpu().long()
        assert torch.equal(gg[clear], mul[g, g0][clear]), g         # the element follows the input: h' = g h
        # the same picture, resampled through a different element: white noise times one ulp of a normalised coordinate
        # (tests/test_gpu_parity.py's pixel bound at this frame size), nothing like the O(1) of a wrong element
        d = (yg - y0)[clear]
        assert d.abs().max().item() <= 3e-4 and d.pow(2).mean().sqrt().item() <= 3e-5, (g, d.abs().max().item())
