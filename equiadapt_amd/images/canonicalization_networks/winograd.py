"""Winograd F(2x2, 5x5) convolution for the 5x5 regular->regular layers of the canonicalization network (inference).

Direct convolution spends 25 multiplies per output and channel pair; Cook-Toom at the points {0, 1, -1, 2, -2, inf}
needs 36 per 2x2 output tile = 9 per output.  Pipeline (channels-last, fp32 throughout):

    x (B,H,W,Cin) --eqa_winograd_f2k5_input-->  V (36, tiles, Cin)
    V[xi] @ U[xi]  (strided-batched fp32 GEMM, library)         ->  M (36, tiles, Cout)
    M --eqa_winograd_f2k5_output--> y (B,H-4,W-4,Cout) = [relu](A^T M A + bias)

U = G g G^T is computed once per weight version in fp64 (G carries all the fractions, B^T is integer), so the only extra
rounding at run time is the two small-integer transforms.  Exact in exact arithmetic; in fp32 the result differs from
the direct convolution by ~1e-6 relative (tests/test_gpu_parity.py::test_winograd_conv_matches_direct).
Images are processed in chunks to bound the size of V and M (36/4 x the activation).
"""
import os
from typing import Optional

import torch

from equiadapt_amd import _lib

# scaled filter transform G (6x5): rows for the points 0, 1, -1, 2, -2, inf
_G = torch.tensor([[1 / 4, 0, 0, 0, 0],
                   [1 / 6, 1 / 6, 1 / 6, 1 / 6, 1 / 6],
                   [1 / 6, -1 / 6, 1 / 6, -1 / 6, 1 / 6],
                   [1 / 24, 1 / 12, 1 / 6, 1 / 3, 2 / 3],
                   [1 / 24, -1 / 12, 1 / 6, -1 / 3, 2 / 3],
                   [0, 0, 0, 0, 1]], dtype=torch.float64)

CHUNK_IMAGES = int(os.environ.get("EQA_WINOGRAD_CHUNK", "64"))


def enabled() -> bool:
    return os.environ.get("EQA_WINOGRAD", "1") != "0"


def transform_filters(bank: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, 5, 5) cross-correlation filters -> U (36, Cin, Cout) fp32 = G g G^T, computed in fp64."""
    G = _G.to(bank.device)
    u = torch.einsum("ak,oikl,bl->aboi", G, bank.double(), G)          # (6, 6, Cout, Cin)
    return u.permute(0, 1, 3, 2).reshape(36, bank.shape[1], bank.shape[0]).float().contiguous()


def applicable(x: torch.Tensor, cin: int, cout: int) -> bool:
    H, W = x.shape[-2:]
    return (enabled() and x.is_cuda and x.dtype == torch.float32 and H >= 6 and W >= 6 and (H - 4) % 2 == 0
            and (W - 4) % 2 == 0 and cin >= 32 and cout >= 32 and x.is_contiguous(memory_format=torch.channels_last))


def sums_applicable(x: torch.Tensor, k_next: int) -> bool:
    """Can the output transform emit the next layer's window sums directly (eqa_winograd_f2k5_output_sums)?"""
    OH, OW = x.shape[-2] - 4, x.shape[-1] - 4
    nb = k_next - 1
    return nb in (2, 4) and OH >= 2 * nb + 2 and OW >= 2 * nb + 2


def conv5x5(x: torch.Tensor, U: torch.Tensor, bias: Optional[torch.Tensor], relu: bool,
            in_bias: Optional[torch.Tensor] = None, in_relu: bool = False, sums_k: int = 0) -> torch.Tensor:
    """x: channels-last (B,Cin,H,W) -> channels-last (B,Cout,H-4,W-4) = [relu](conv2d(act(x), g) + bias), g given as U;
    act(x) = [relu](x + in_bias[c]) is applied while the input tiles are loaded (previous layer's epilogue)."""
    lib = _lib.load()
    B, Cin, H, W = x.shape
    Cout = U.shape[2]
    OH, OW = H - 4, W - 4
    TY, TX = OH // 2, OW // 2
    if sums_k:
        # fused tail: return the (B, Cout, k, k) fp64 window sums of the activation instead of the activation itself
        S = torch.empty((B, Cout, sums_k, sums_k), dtype=torch.float64, device=x.device)
        y = None
    else:
        y = torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    stream = torch.cuda.current_stream().cuda_stream
    chunk = min(CHUNK_IMAGES, B)
    V = torch.empty((36, chunk * TY * TX, Cin), dtype=torch.float32, device=x.device)
    M = torch.empty((36, chunk * TY * TX, Cout), dtype=torch.float32, device=x.device)
    xs = x.data_ptr()
    ys = y.data_ptr() if y is not None else 0
    if sums_k:
        ws = torch.empty((max(lib.eqa_winograd_f2k5_output_sums_workspace_bytes(chunk, OH, Cout, sums_k), 4) // 4,),
                         dtype=torch.float32, device=x.device)
    p_bias = bias.data_ptr() if bias is not None else None
    p_in_bias = in_bias.data_ptr() if in_bias is not None else None
    with torch.cuda.device(x.device):
        for b0 in range(0, B, chunk):
            n = min(chunk, B - b0)
            t = n * TY * TX
            st = lib.eqa_winograd_f2k5_input(xs + b0 * H * W * Cin * 4, V.data_ptr(), p_in_bias, int(in_relu), n, H, W, Cin, stream)
            _lib.check(st, "eqa_winograd_f2k5_input")
            if n == chunk:
                torch.bmm(V, U, out=M)
            else:  # last, smaller chunk: the kernels address (36, t, C) densely
                Vn, Mn = V.view(-1)[: 36 * t * Cin].view(36, t, Cin), M.view(-1)[: 36 * t * Cout].view(36, t, Cout)
                torch.bmm(Vn, U, out=Mn)
            if sums_k:
                st = lib.eqa_winograd_f2k5_output_sums(M.data_ptr(), p_bias, int(relu), S.data_ptr() + b0 * Cout * sums_k * sums_k * 8,
                                                       ws.data_ptr(), n, OH, OW, Cout, sums_k, stream)
                _lib.check(st, "eqa_winograd_f2k5_output_sums")
            else:
                st = lib.eqa_winograd_f2k5_output(M.data_ptr(), p_bias, int(relu), ys + b0 * OH * OW * Cout * 4, n, OH, OW, Cout, stream)
                _lib.check(st, "eqa_winograd_f2k5_output")
    return S if sums_k else y
