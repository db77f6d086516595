"""Winograd F(m x m, 5x5) convolution for the 5x5 regular->regular layers of the canonicalization network (inference).

Direct convolution spends 25 multiplies per output and channel pair.  Cook-Toom with N = m + 4 points per axis needs
N*N per m x m output tile:  m = 2, points {0, 1, -1, 2, -2, inf}: 36/4 = 9 per output;  m = 4, points
{0, 1, -1, 2, -2, 1/2, -1/2, inf}: 64/16 = 4 per output.  Pipeline (channels-last, fp32 throughout), P = N*N:

    x (B,H,W,Cin) --eqa_winograd_f{m}k5_input-->  V (tiles, P, Cin)
    V[:, a] @ U[a]  (eqa_plane_gemm: hand-written batched fp32-MFMA GEMM; channel counts off the 32-multiples: the
                     framework's strided-batched GEMM)               ->  M (tiles, P, Cout)
    M --eqa_winograd_f{m}k5_output--> y (B,H-4,W-4,Cout) = [relu](A^T M A + bias)

U = G g G^T is computed once per weight version in fp64 (G carries the non-dyadic fractions; B^T and A^T are exact in
fp32), so the only extra rounding at run time is in the two transforms and the P-plane products.  Exact in exact
arithmetic (tests/test_abi_and_host.py checks the matrices below in rational arithmetic); in fp32, against an fp64
convolution with 256 channels: m = 2: 7e-6 of max|y|, m = 4: 9e-6 (direct fp32: 3e-7) --
tests/test_gpu_parity.py::test_winograd_conv_matches_direct bounds both by 2e-5.
m = 4 is used whenever 4 divides H-4 and W-4, else m = 2.  Images are processed in chunks to bound the size of V and M.
"""
import os
from fractions import Fraction
from typing import List, Optional, Sequence, Tuple

import torch

from equiadapt_amd import _lib, ops
from equiadapt_amd.ops import _timed

# interpolation points (the last, implicit one is infinity)
POINTS = {2: (0, 1, -1, 2, -2), 4: (0, 1, -1, 2, -2, Fraction(1, 2), Fraction(-1, 2))}

CHUNK_IMAGES = int(os.environ.get("EQA_WINOGRAD_CHUNK", "64"))
KEEP_V_FOR_BACKWARD = os.environ.get("EQA_WINOGRAD_KEEP_V", "1") != "0"
PLANE_GEMM = os.environ.get("EQA_WINOGRAD_GEMM", "own") != "lib"   # "lib": the framework's strided-batched GEMM (A/B runs)

_packed_cache: dict = {}


def _packed_filters(U: torch.Tensor) -> torch.Tensor:
    """U in the operand-fragment order of eqa_plane_gemm, cached per (storage, version): inference passes the same cached U every
    call; training builds a new U per step and pays one 64 x Cin x Cout copy for it."""
    key = (U.data_ptr(), U._version, tuple(U.shape), str(U.device))
    hit = _packed_cache.get(key)
    if hit is None:
        if len(_packed_cache) >= 16:
            _packed_cache.clear()
        hit = (U, ops.pack_plane_gemm_weights(U))      # (keeps U alive so that its address cannot be reused under this key)
        _packed_cache[key] = hit
    return hit[1]


def _polymul(a: Sequence[Fraction], b: Sequence[Fraction]) -> List[Fraction]:
    out = [Fraction(0)] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] += x * y
    return out


def cook_toom(m: int, r: int = 5) -> Tuple[list, list, list]:
    """(A^T, G, B^T) as rational matrices for the correlation y_i = sum_j d_{i+j} g_j, i < m:
    y = A^T [(G g) * (B^T d)].  Row k of B^T holds the coefficients of prod_{j != k} (x - p_j) (last row: all points),
    row k of G the powers of p_k over prod_{j != k} (p_k - p_j) -- the kernels' wino_bt / wino_at hard-code B^T and A^T
    with these signs except rows 1, 2 of m = 2, which are negated in both B^T and G (see `g_matrix`)."""
    pts = [Fraction(p) for p in POINTS[m]]
    n = m + r - 1
    assert len(pts) == n - 1
    full = [Fraction(1)]
    for p in pts:
        full = _polymul(full, [-p, Fraction(1)])
    bt, g = [], []
    for k, p in enumerate(pts):
        q, den = [Fraction(1)], Fraction(1)
        for j, pj in enumerate(pts):
            if j != k:
                q = _polymul(q, [-pj, Fraction(1)])
                den *= p - pj
        bt.append(q + [Fraction(0)] * (n - len(q)))
        g.append([p ** j / den for j in range(r)])
    bt.append(full)
    g.append([Fraction(0)] * (r - 1) + [Fraction(1)])
    at = [[p ** i for p in pts] + [Fraction(1 if i == m - 1 else 0)] for i in range(m)]
    return at, g, bt


def kernel_bt_signs(m: int) -> List[int]:
    """Sign of the kernels' B^T rows relative to `cook_toom` (wino_bt in csrc/winograd.hip)."""
    return [1, -1, -1, 1, 1, 1] if m == 2 else [1] * 8


def g_matrix(m: int) -> torch.Tensor:
    """(N, 5) fp64 filter transform matching the kernels' B^T."""
    _, g, _ = cook_toom(m)
    sg = kernel_bt_signs(m)
    return torch.tensor([[float(sg[k] * v) for v in row] for k, row in enumerate(g)], dtype=torch.float64)


def enabled() -> bool:
    return os.environ.get("EQA_WINOGRAD", "1") != "0"


def tile_for(x: torch.Tensor) -> int:
    """Output tile size m for an input of this spatial size (EQA_WINOGRAD_TILE=2 forces the small tile)."""
    H, W = x.shape[-2:]
    if (H - 4) % 4 == 0 and (W - 4) % 4 == 0 and H >= 8 and W >= 8 and os.environ.get("EQA_WINOGRAD_TILE", "4") != "2":
        return 4
    return 2


def transform_filters(bank: torch.Tensor, m: int = 2) -> torch.Tensor:
    """(Cout, Cin, 5, 5) cross-correlation filters -> U (N*N, Cin, Cout) fp32 = G g G^T, computed in fp64."""
    G = g_matrix(m).to(bank.device)
    n = m + 4
    u = torch.einsum("ak,oikl,bl->aboi", G, bank.double(), G)          # (N, N, Cout, Cin)
    return u.permute(0, 1, 3, 2).reshape(n * n, bank.shape[1], bank.shape[0]).float().contiguous()


def applicable(x: torch.Tensor, cin: int, cout: int) -> bool:
    H, W = x.shape[-2:]
    return (enabled() and x.is_cuda and x.dtype == torch.float32 and H >= 6 and W >= 6 and (H - 4) % 2 == 0
            and (W - 4) % 2 == 0 and cin >= 32 and cout >= 32 and x.is_contiguous(memory_format=torch.channels_last))


def sums_applicable(x: torch.Tensor, k_next: int, m: int = 2) -> bool:
    """Can the output transform emit the next layer's window sums directly (eqa_winograd_f{m}k5_output_sums)?"""
    OH, OW = x.shape[-2] - 4, x.shape[-1] - 4
    nb = k_next - 1
    return nb in (2, 4) and nb % m == 0 and OH >= 2 * nb + m and OW >= 2 * nb + m


def filter_grad(x: torch.Tensor, dY: torch.Tensor, m: int, V_saved: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dL/dU (P, Cin, Cout) of y = conv5x5(x, U):  dU[a] = V[:, a]^T dM[:, a]  with V = B^T x B (input transform) and
    dM = A dY A^T (eqa_winograd_f{m}k5_output_adjoint); x, dY channels-last (B,Cin,H,W) / (B,Cout,H-4,W-4)."""
    lib = _lib.load()
    B, Cin, H, W = x.shape
    Cout = dY.shape[1]
    OH, OW = H - 4, W - 4
    n = m + 4
    P = n * n
    TY, TX = OH // m, OW // m
    f_in, f_adj = getattr(lib, f"eqa_winograd_f{m}k5_input"), getattr(lib, f"eqa_winograd_f{m}k5_output_adjoint")
    chunk = min(CHUNK_IMAGES * (m * m // 4), B)
    have_V = V_saved is not None and chunk == B
    V = V_saved if have_V else torch.empty((chunk * TY * TX, P, Cin), dtype=torch.float32, device=x.device)
    dM = torch.empty((chunk * TY * TX, P, Cout), dtype=torch.float32, device=x.device)
    dU = torch.zeros((P, Cin, Cout), dtype=torch.float32, device=x.device)
    stream = torch.cuda.current_stream().cuda_stream
    with torch.cuda.device(x.device):
        for b0 in range(0, B, chunk):
            nimg = min(chunk, B - b0)
            t = nimg * TY * TX
            if not have_V:
                _lib.check(f_in(x.data_ptr() + b0 * H * W * Cin * 4, V.data_ptr(), None, 0, nimg, H, W, Cin, stream), "winograd input")
            _lib.check(f_adj(dY.data_ptr() + b0 * OH * OW * Cout * 4, dM.data_ptr(), nimg, OH, OW, Cout, stream), "winograd output adjoint")
            # (P, Cin, t) x (P, t, Cout): both operands are strided views of the tile-major buffers, no copies
            dU.baddbmm_(V[:t].permute(1, 2, 0), dM[:t].permute(1, 0, 2))
    return dU


class Conv5x5Function(torch.autograd.Function):
    """y = conv2d(x, bank) (5x5, stride 1, no padding, channels-last) through Winograd, with both gradients:
    dx = conv5x5(zero-pad(dy, 4), flipped + transposed bank)  -- the same kernels --, and
    dbank = G^T dU G with dU from `filter_grad`.  Training counterpart of the inference path in escnn_networks.py."""

    @staticmethod
    def forward(ctx, x, bank, m, with_stats=False):
        """with_stats (the caller has checked ``stats_supported``): also return the fp64 partial sums (rows, Cout, 2) of the
        output's per-channel sum / sum of squares, taken by the FFT convolution's inverse transform on the way out."""
        from equiadapt_amd.images.canonicalization_networks import fftconv

        keep: list = []
        stats: list = []
        ctx.fft = fftconv.TRAIN_FORWARD and fftconv.applicable(x, bank.shape[1], bank.shape[0])
        if ctx.fft:
            # forward pass and both gradients as FFT convolutions (2.5 instead of 4 multiplies per output; the filter spectra
            # are rebuilt by one kernel, the input spectra are kept for the filter gradient)
            y = fftconv.conv5x5(x, fftconv.spectra_for(bank.detach()), None, False, keep_V=keep, stats=stats if with_stats else None)
        else:
            assert not with_stats
            y = conv5x5(x, transform_filters(bank.detach(), m), None, False, keep_V=keep if KEEP_V_FOR_BACKWARD else None)
        ctx.save_for_backward(x, bank, *keep)     # V is kept: HBM is 288 GB, recomputing it is a 1-2 ms pass
        ctx.m = m
        if with_stats:
            ctx.mark_non_differentiable(stats[0])
            return y, stats[0]
        return y

    @staticmethod
    def stats_supported(x, bank) -> bool:
        from equiadapt_amd.images.canonicalization_networks import fftconv

        return bool(fftconv.TRAIN_FORWARD and fftconv.applicable(x, bank.shape[1], bank.shape[0])
                    and fftconv.output_stats_supported(x.shape[0], x.shape[2] - 4, x.shape[3] - 4, bank.shape[0]))

    @staticmethod
    def backward(ctx, dy, *_):
        x, bank, *keep = ctx.saved_tensors
        m = ctx.m
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = dbank = None
        if ctx.fft:
            # both gradients in the frequency domain, from one transform of the output gradient's 44 x 44 tiles
            from equiadapt_amd.images.canonicalization_networks import fftconv

            G = fftconv.grad_spectra(dy)
            if ctx.needs_input_grad[0]:
                dx = fftconv.input_grad(dy, bank, G)
            if ctx.needs_input_grad[1]:
                dbank = fftconv.filter_grad(keep[0], dy, bank.shape[1], G).to(bank.dtype)
            return dx, dbank, None, None
        if ctx.needs_input_grad[0]:
            bank_t = bank.detach().flip(-1, -2).transpose(0, 1).contiguous()       # (Cin, Cout, 5, 5)
            dx = conv5x5(dy, transform_filters(bank_t, m), None, False, pad=4)   # zero padding inside the input transform
        if ctx.needs_input_grad[1]:
            n = m + 4
            G = g_matrix(m).to(dy.device)
            dU = filter_grad(x, dy, m, keep[0] if keep else None).double().view(n, n, bank.shape[1], bank.shape[0])
            dbank = torch.einsum("ak,bl,abio->oikl", G, G, dU).to(bank.dtype)
        return dx, dbank, None, None


def conv5x5(x: torch.Tensor, U: torch.Tensor, bias: Optional[torch.Tensor], relu: bool,
            in_bias: Optional[torch.Tensor] = None, in_relu: bool = False, sums_k: int = 0, pad: int = 0,
            keep_V: Optional[list] = None) -> torch.Tensor:
    """x: channels-last (B,Cin,H,W) -> channels-last (B,Cout,H-4,W-4) = [relu](conv2d(act(x), g) + bias), g given as
    U = transform_filters(g, m) (m is read off U's plane count); act(x) = [relu](x + in_bias[c]) is applied while the
    input tiles are loaded (previous layer's epilogue).  ``pad``: convolve x zero-padded by that many pixels (no padded copy
    is made).  ``keep_V``: a list that receives the transformed input V when the batch is one chunk (training: the filter
    gradient reuses it instead of transforming x again)."""
    lib = _lib.load()
    B, Cin, H, W = x.shape
    xp_in, (H, W) = (H, W), (H + 2 * pad, W + 2 * pad)   # from here on H, W are the logical (padded) sizes
    P, _, Cout = U.shape
    m = {36: 2, 64: 4}[P]
    f_in, f_out, f_sums = (getattr(lib, f"eqa_winograd_f{m}k5_{n}") for n in ("input", "output", "output_sums"))
    f_in_pad = getattr(lib, f"eqa_winograd_f{m}k5_input_padded")
    OH, OW = H - 4, W - 4
    TY, TX = OH // m, OW // m
    if sums_k:
        # fused tail: return the (B, Cout, k, k) fp64 window sums of the activation instead of the activation itself
        S = torch.empty((B, Cout, sums_k, sums_k), dtype=torch.float64, device=x.device)
        y = None
    else:
        y = torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    stream = torch.cuda.current_stream().cuda_stream
    chunk = min(CHUNK_IMAGES * (m * m // 4), B)       # same V / M footprint per chunk for both tile sizes
    V = torch.empty((chunk * TY * TX, P, Cin), dtype=torch.float32, device=x.device)
    M = torch.empty((chunk * TY * TX, P, Cout), dtype=torch.float32, device=x.device)
    xs = x.data_ptr()
    ys = y.data_ptr() if y is not None else 0
    if sums_k:
        ws = torch.empty((max(lib.eqa_winograd_f2k5_output_sums_workspace_bytes(chunk, OH, Cout, sums_k), 4) // 4,),
                         dtype=torch.float32, device=x.device)
    p_bias = bias.data_ptr() if bias is not None else None
    p_in_bias = in_bias.data_ptr() if in_bias is not None else None
    Upk = _packed_filters(U) if PLANE_GEMM and ops.plane_gemm_supported(Cin, Cout) else None
    with torch.cuda.device(x.device):
        for b0 in range(0, B, chunk):
            n = min(chunk, B - b0)
            t = n * TY * TX
            with _timed("winograd_input"):
                if pad:
                    st = f_in_pad(xs + b0 * xp_in[0] * xp_in[1] * Cin * 4, V.data_ptr(), n, xp_in[0], xp_in[1], Cin, pad, stream)
                else:
                    st = f_in(xs + b0 * H * W * Cin * 4, V.data_ptr(), p_in_bias, int(in_relu), n, H, W, Cin, stream)
            _lib.check(st, f"eqa_winograd_f{m}k5_input")
            # plane a is the strided matrix V[:, a, :] (row stride P*Cin): no copy, the library takes lda / batch stride
            with _timed("winograd_gemm"):
                if Upk is not None:
                    ops.plane_gemm(V, Upk, M, t)      # hand-written fp32-MFMA batched GEMM (csrc/planegemm.hip)
                else:
                    torch.bmm(V[:t].permute(1, 0, 2), U, out=M[:t].permute(1, 0, 2))
            if sums_k:
                with _timed("winograd_output_sums"):
                    st = f_sums(M.data_ptr(), p_bias, int(relu), S.data_ptr() + b0 * Cout * sums_k * sums_k * 8,
                                ws.data_ptr(), n, OH, OW, Cout, sums_k, stream)
                _lib.check(st, f"eqa_winograd_f{m}k5_output_sums")
            else:
                with _timed("winograd_output"):
                    st = f_out(M.data_ptr(), p_bias, int(relu), ys + b0 * OH * OW * Cout * 4, n, OH, OW, Cout, stream)
                _lib.check(st, f"eqa_winograd_f{m}k5_output")
    if keep_V is not None and chunk == B:
        keep_V.append(V)
    return S if sums_k else y
