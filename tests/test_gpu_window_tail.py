"""Training counterparts of the linearised last layer (last convolution + group mean = window sums + a GEMV; reference:
escnn_networks.py:93-117, custom_equivariant_networks.py:80-93): the window sums' backward as a class table
(eqa_window_grad_table) and the GEMV's backward (eqa_window_sums_gemv_bwd) against the fp64 torch forms they replace."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    from equiadapt_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,C,H,W,k", [(7, 64, 48, 48, 9), (3, 256, 88, 88, 5), (5, 24, 19, 31, 3), (2, 100, 21, 20, 10), (4, 8, 9, 9, 5), (1, 4, 5, 7, 1)])
def test_window_grad_table_equals_the_mask_einsum(dev, B, C, H, W, k, monkeypatch):
    from equiadapt_amd import ops
    from equiadapt_amd.images.canonicalization_networks import escnn_networks as en
    from equiadapt_amd.images.canonicalization_networks.pooling import WindowSumsFunction

    dS = torch.randn(B, C, k, k, dtype=torch.float64, generator=torch.Generator().manual_seed(B * C)).to(dev)
    got = ops.window_grad_table(dS, H, W)
    monkeypatch.setenv("EQA_WS_TABLE_KERNEL", "0")
    want = en._window_grad_table(dS, H, W, k)                       # the einsum form
    assert got.shape == want.shape == (B, 2 * k - 1, 2 * k - 1, C)
    assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    # and through the autograd function, against autograd through an explicit unfold of the definition (fp64)
    if C % 4 == 0 and k > 1:
        x = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(3)).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        S = WindowSumsFunction.apply(x, k)
        (S * dS).sum().backward()
        x64 = x.detach().double().requires_grad_(True)
        S64 = torch.stack([torch.stack([x64[:, :, u:u + H - k + 1, v:v + W - k + 1].sum(dim=(2, 3)) for v in range(k)], -1) for u in range(k)], -2)
        assert (S.detach() - S64.detach()).abs().max().item() <= 1e-4 * S64.abs().max().item()
        (S64 * dS).sum().backward()
        assert (x.grad.double() - x64.grad).abs().max().item() <= 1e-5 * x64.grad.abs().max().item()


@pytest.mark.parametrize("B,K,E", [(512, 5184, 4), (256, 6400, 8), (33, 75, 16), (1, 800, 8), (700, 200, 3)])
def test_window_sums_linear_function_matches_fp64_autograd(dev, B, K, E):
    from equiadapt_amd.images.canonicalization_networks.pooling import WindowSumsLinearFn

    g = torch.Generator().manual_seed(K + E)
    S = (torch.randn(B, K, dtype=torch.float64, generator=g) * 30).to(dev).requires_grad_(True)
    W = torch.randn(E, K, dtype=torch.float64, generator=g).to(dev).requires_grad_(True)
    up = torch.randn(B, E, generator=g).to(dev)
    scale = 1.0 / 1234.0
    act = WindowSumsLinearFn.apply(S, W, scale)
    assert act.dtype == torch.float32 and act.shape == (B, E)
    (act * up).sum().backward()
    S2, W2 = S.detach().clone().requires_grad_(True), W.detach().clone().requires_grad_(True)
    want = S2 @ W2.t() * scale
    (want * up.double()).sum().backward()
    assert (act.double() - want).abs().max().item() <= 2e-6 * want.abs().max().item()
    assert (S.grad - S2.grad).abs().max().item() <= 1e-12 * S2.grad.abs().max().item() + 1e-300
    assert (W.grad - W2.grad).abs().max().item() <= 1e-12 * W2.grad.abs().max().item()
    again = WindowSumsLinearFn.apply(S.detach().requires_grad_(True), W.detach().requires_grad_(True), scale)
    assert torch.equal(again, act)


def test_window_tail_kernels_edge_cases(dev):
    from equiadapt_amd import _lib, ops

    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    assert ops.window_grad_table(torch.empty(0, 8, 5, 5, dtype=torch.float64, device=dev), 20, 20).shape == (0, 9, 9, 8)
    dS = torch.zeros(2, 8, 5, 5, dtype=torch.float64, device=dev)
    tab = torch.empty(2, 9, 9, 8, device=dev)
    assert lib.eqa_window_grad_table(dS.data_ptr(), tab.data_ptr(), 2, 8, 8, 20, 5, st) == -3       # H < 2 (k - 1) + 1: classes overlap
    assert lib.eqa_window_grad_table(dS.data_ptr(), tab.data_ptr(), 2, 8, 4, 20, 5, st) == -1       # H < k
    assert lib.eqa_window_grad_table(dS.data_ptr(), tab.data_ptr(), 2, 8, 40, 40, 11, st) == -3     # k beyond the window-sum kernels
    S = torch.empty(0, 100, dtype=torch.float64, device=dev)
    W = torch.randn(4, 100, dtype=torch.float64, device=dev)
    dS0, dW0 = ops.window_sums_gemv_bwd(torch.empty(0, 4, device=dev), W, S, 0.5, True, True)
    assert dS0.shape == (0, 100) and (dW0 == 0).all()
    assert lib.eqa_window_sums_gemv_bwd(None, W.data_ptr(), S.data_ptr(), None, None, None, 3, 100, 17, 1.0, st) == -3     # E > 16
    assert lib.eqa_window_sums_gemv_bwd_workspace_bytes(100, 4) == 16 * 4 * 100 * 8
