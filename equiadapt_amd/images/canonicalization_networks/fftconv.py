"""5x5 stride-1 convolutions as an overlap-save FFT convolution on 48x48 tiles: the regular -> regular layers of the
canonicalization network (reference: equiadapt/images/canonicalization_networks/escnn_networks.py:67-91) in inference
(`conv5x5`) and in training (`conv5x5` with ``keep_V``, `grad_spectra`, `filter_grad`, `input_grad`; used by
`winograd.Conv5x5Function`).

48x48 real FFT tiles produce 44x44 outputs each; per frequency the channel contraction is a complex matrix product that the
GEMM library runs as a real one (see include/eqa_hip.h, eqa_fft48k5_*).  Where the tiles fit the output (88 = 2 x 44 at the
headline shape) this needs 2.5 real multiplies per output against 4 for Winograd F(4x4,5x5), and spectra of 1.24x the
activation size against 4x.  fp32 throughout."""
import math
import os
from typing import Optional

import torch

from equiadapt_amd import _lib, ops
from equiadapt_amd.ops import _timed

N, NH, OUT = 48, 25, 44
F = N * (NH - 2) + 2 * NH     # 1154 stored frequencies: the conjugate-redundant half of the kx = 0 / 24 columns is dropped
ENABLED = os.environ.get("EQA_CONV5_FFT", "1") != "0"
# The filter spectra (1154 x 2Cin x 2Cout floats: 1.21 GB at 256 channels) are streamed once per call, the Winograd filters are
# 17 MB: below ~32 tiles (8 images of 92 x 92) the Winograd path is faster (measured: B=4 0.29 vs 0.37 ms, B=8 0.48 vs 0.46 ms).
MIN_TILES = int(os.environ.get("EQA_FFT_MIN_TILES", "32"))
TRAIN_FORWARD = os.environ.get("EQA_FFT_TRAIN", "1") != "0"     # forward pass of winograd.Conv5x5Function through this path
# The per-frequency channel contraction: "3m" = the hand-written complex GEMM on the fp32 MFMA in the 3-multiplication form
# (eqa_fft48k5_cgemm3m; channel counts it takes: Cin % 32 == 0, Cout % 64 == 0), "lib" = the library's real batched GEMM.
GEMM = os.environ.get("EQA_FFT_GEMM", "3m")
# How the hand-written GEMM multiplies: "f32" = the fp32 matrix instruction; "9" / "6" = every fp32 operand split exactly into three
# bf16 pieces and that many piece products on the bf16 matrix cores (eqa_fft48k5_cgemm3m_bf16x3: 9 = every product, the same
# exact-products / fp32-accumulation contract as the fp32 instruction; 6 = without the three products of relative size <= 2^-24).
# "auto" (default since round 6): six piece products on the shapes where they are BOTH faster and no further from an fp64
# evaluation than the fp32 instruction -- Cin >= AUTO_MIN_CIN (the piece form accumulates 8 / 16 products per rounding, the fp32
# instruction 2: its advantage grows with the contraction length; at Cin = 32 it is 7-11 % further away, profiles/r05/
# kbench_gemm_error.txt) and Cout % 128 == 0 (the block form: A split once per block, 1.5 x the fp32 form's speed) -- the fp32
# instruction everywhere else.  The rule is gated by tests/test_gpu_parity.py::test_auto_gemm_form_is_no_further_from_fp64.
GEMM_PIECES = os.environ.get("EQA_FFT_GEMM_PIECES", "auto")
AUTO_MIN_CIN = int(os.environ.get("EQA_FFT_GEMM_AUTO_MIN_CIN", "128"))
# "h3" (round 6, the default where it applies): TWO fp16 pieces per operand, three exact products per product on the fp16 matrix
# cores (eqa_fft48k5_cgemm3m_f16x2).  Half the matrix instructions of "6" and closer to fp64 than "6" AND the fp32 instruction for
# Cin >= AUTO_MIN_CIN_H3 (fewer accumulator roundings; profiles/r06/kbench_gemm_error.txt) -- but fp16's range wants the operands
# scaled, so it runs only where the producer of V hands over a bound of |V| on the device: the fused lifting kernel on non-negative
# (relu) activations, whose DC bins bound every other bin (eqa_lift5_fft48k5_input_dcmax).  Everywhere else "auto" stays as above.
AUTO_MIN_CIN_H3 = int(os.environ.get("EQA_FFT_GEMM_AUTO_MIN_CIN_H3", "64"))
DCMAX_SLOTS = 256       # EQA_LIFT5_DCMAX_SLOTS (include/eqa_hip.h)
LAST_FORM = None        # the form the most recent `contract` on the hand-written GEMM ran in (bench.py reports it)


def f16_form_takes(cin: int, cout: int) -> bool:
    return cout % 128 == 0 and cin % 32 == 0


def gemm_form(cin: int, cout: int, bounded: bool = False) -> str:
    """"f32" / "9" / "6" / "h3": how `contract` multiplies a (.., 2 cin) . (2 cin, 2 cout) complex product (see GEMM_PIECES).
    ``bounded``: the caller holds an upper bound of |V| on the device (what the fp16 form "h3" needs)."""
    if GEMM_PIECES in ("f32", "9", "6"):
        return GEMM_PIECES
    if bounded and f16_form_takes(cin, cout) and (GEMM_PIECES == "h3" or cin >= AUTO_MIN_CIN_H3):
        return "h3"
    if GEMM_PIECES == "h3":        # no bound, or a shape the fp16 kernel does not take: as "auto" without it
        return "6" if cin >= AUTO_MIN_CIN and cout % 128 == 0 else "f32"
    return "6" if cin >= AUTO_MIN_CIN and cout % 128 == 0 else "f32"


class Spectra3M:
    """Filter spectra in the operand order of eqa_fft48k5_cgemm3m: ``data`` is the flat (F * Cin * Cout * 3) fp32 buffer
    [Br | Bi | Br + Bi] per (frequency, K-stage, 32 output channels); ``pieces()``: the same split into three bf16 pieces per value
    in the fragment order of the bf16 matrix instruction (built on first use)."""

    __slots__ = ("data", "cin", "cout", "_pieces", "_pieces_f16")

    def __init__(self, data: torch.Tensor, cin: int, cout: int):
        self.data, self.cin, self.cout, self._pieces, self._pieces_f16 = data, cin, cout, None, None

    def pieces_f16(self):
        """(Bh, b_scale): the spectra times the power of two b_scale (max |.| -> at most 2^14) as two fp16 pieces per value, in the
        fragment order of the fp16 matrix instruction (built on first use: one host synchronisation for the maximum)."""
        if self._pieces_f16 is None:
            lib = _lib.load()
            mx = float(self.data.abs().max().item())
            ex = 14 - math.frexp(mx)[1] if 0.0 < mx < float("inf") else 0        # mx <= 2^frexp(mx)[1]
            scale = math.ldexp(1.0, max(-100, min(100, ex)))
            bh = torch.empty(lib.eqa_fft48k5_spectra3m_f16_bytes(self.cin, self.cout) // 2, dtype=torch.int16, device=self.data.device)
            with torch.cuda.device(self.data.device):
                _lib.check(lib.eqa_fft48k5_spectra3m_split_f16(self.data.data_ptr(), bh.data_ptr(), self.cin, self.cout, scale,
                                                               torch.cuda.current_stream().cuda_stream), "eqa_fft48k5_spectra3m_split_f16")
            self._pieces_f16 = (bh, scale)
        return self._pieces_f16

    def pieces(self) -> torch.Tensor:
        if self._pieces is None:
            lib = _lib.load()
            bp = torch.empty(lib.eqa_fft48k5_spectra3m_bf16_bytes(self.cin, self.cout) // 2, dtype=torch.int16, device=self.data.device)
            with torch.cuda.device(self.data.device):
                _lib.check(lib.eqa_fft48k5_spectra3m_split(self.data.data_ptr(), bp.data_ptr(), self.cin, self.cout,
                                                           torch.cuda.current_stream().cuda_stream), "eqa_fft48k5_spectra3m_split")
            self._pieces = bp
        return self._pieces


def gemm3m_supported(cin: int, cout: int) -> bool:
    return GEMM == "3m" and bool(_lib.load().eqa_fft48k5_cgemm3m_supported(cin, cout))


def filter_spectra3m(bank: torch.Tensor, correlate: bool = True) -> Spectra3M:
    """(Cout, Cin, 5, 5) device bank -> the same spectra as `filter_spectra`, laid out for the 3-multiplication complex GEMM."""
    lib = _lib.load()
    Cout, Cin = bank.shape[:2]
    if not (bank.is_cuda and bank.dtype == torch.float32 and lib.eqa_fft48k5_cgemm3m_supported(Cin, Cout)):
        raise RuntimeError("filter_spectra3m: fp32 device bank with Cin % 32 == 0 and Cout % 64 == 0 expected")
    B3 = torch.empty(lib.eqa_fft48k5_spectra3m_floats(Cin, Cout), dtype=torch.float32, device=bank.device)
    with torch.cuda.device(bank.device):
        _lib.check(lib.eqa_fft48k5_filter_spectra3m(bank.contiguous().data_ptr(), B3.data_ptr(), Cout, Cin, int(correlate),
                                                    torch.cuda.current_stream().cuda_stream), "eqa_fft48k5_filter_spectra3m")
    return Spectra3M(B3, Cin, Cout)


def spectra_for(bank: torch.Tensor, correlate: bool = True):
    """The filter spectra in the form the contraction will consume: `Spectra3M` where the hand-written GEMM takes the channel
    counts, else the (F, 2Cin, 2Cout) real form for the library GEMM."""
    Cout, Cin = bank.shape[:2]
    if bank.is_cuda and bank.dtype == torch.float32 and gemm3m_supported(Cin, Cout):
        return filter_spectra3m(bank, correlate)
    return filter_spectra(bank, correlate=correlate)


def contract(V: torch.Tensor, B, M: int, vbound: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Mo[f] = V[f] . B[f] for every stored frequency: V (F, M, 2Cin) view of a pitched buffer -> Mo (F, M, 2Cout) likewise.
    ``vbound``: a device fp32 tensor whose maximum bounds every |Re|, |Im| of V (the producer's DC bins) -- admits the fp16 form."""
    dev = V.device
    if isinstance(B, Spectra3M):
        lib = _lib.load()
        assert V.shape[2] == 2 * B.cin and V.stride(1) == 2 * B.cin and V.stride(0) == lib.eqa_fft48k5_tile_pitch(M) * 2 * B.cin
        Mo = spectra_buffer(M, 2 * B.cout, dev)
        global LAST_FORM
        form = LAST_FORM = gemm_form(B.cin, B.cout, vbound is not None)
        if form == "h3":
            bh, b_scale = B.pieces_f16()
            _lib.check(lib.eqa_fft48k5_cgemm3m_f16x2(V.data_ptr(), bh.data_ptr(), Mo.data_ptr(), M, B.cin, B.cout, vbound.data_ptr(),
                                                     vbound.numel(), b_scale, torch.cuda.current_stream().cuda_stream),
                       "eqa_fft48k5_cgemm3m_f16x2")
            return Mo
        if form in ("9", "6"):
            _lib.check(lib.eqa_fft48k5_cgemm3m_bf16x3(V.data_ptr(), B.pieces().data_ptr(), Mo.data_ptr(), M, B.cin, B.cout, int(form),
                                                      torch.cuda.current_stream().cuda_stream), "eqa_fft48k5_cgemm3m_bf16x3")
            return Mo
        _lib.check(lib.eqa_fft48k5_cgemm3m(V.data_ptr(), B.data.data_ptr(), Mo.data_ptr(), M, B.cin, B.cout,
                                           torch.cuda.current_stream().cuda_stream), "eqa_fft48k5_cgemm3m")
        return Mo
    return torch.bmm(V, B, out=spectra_buffer(M, B.shape[2], dev))


def freq_index():
    """(ky, kx) of stored frequency f, in the kernels' order: f = 23 ky + kx - 1 for 0 < kx < 24 (all 48 ky), then
    1104 + 2 ky + (kx == 24) for the edge columns, ky <= 24."""
    ky = torch.arange(N).repeat_interleave(NH - 2)
    kx = torch.arange(1, NH - 1).repeat(N)
    eky = torch.arange(NH).repeat_interleave(2)
    ekx = torch.tensor([0, NH - 1]).repeat(NH)
    return torch.cat([ky, eky]), torch.cat([kx, ekx])


def tiles(n: int) -> int:
    return 0 if n <= 4 else (n - 4 + OUT - 1) // OUT


class GroupedMap:
    """An activation map in the channel-group-major layout (nimg, C/16, H, W, 16) written by `ops.lift_conv_grouped` and read by
    `conv5x5` (eqa_fft48k5_input_grouped): `.shape` is the logical (nimg, C, H, W)."""

    def __init__(self, data: torch.Tensor):
        assert data.dim() == 5 and data.shape[-1] == 16 and data.is_contiguous()
        self.data = data
        self.shape = torch.Size((data.shape[0], data.shape[1] * 16, data.shape[2], data.shape[3]))
        self.device, self.dtype, self.is_cuda = data.device, data.dtype, data.is_cuda

    def data_ptr(self) -> int:
        return self.data.data_ptr()

    def to_channels_last(self) -> torch.Tensor:
        """The same map as an ordinary channels-last (nimg, C, H, W) tensor (a copy)."""
        n, g, h, w, _ = self.data.shape
        return self.data.permute(0, 1, 4, 2, 3).reshape(n, g * 16, h, w).contiguous(memory_format=torch.channels_last)


class LiftedInput:
    """A lifting layer that has NOT run: (x channels-last (nimg, 3, H0, W0), folded 5 x 5 bank (C, 3, 5, 5) channels-last, bias (C),
    relu) standing for the map [relu](conv2d(x, bank) + bias) of logical `.shape` (nimg, C, H0 - 4, W0 - 4).  `conv5x5` takes it in
    place of that map and produces the map's tile spectra straight from x (eqa_lift5_fft48k5_input: the lifting convolution fused
    into the forward FFT-48 transform -- the 2.2 GB map of the headline shape is neither written nor read)."""

    def __init__(self, x: torch.Tensor, bank: torch.Tensor, bias: Optional[torch.Tensor], relu: bool, pieces: Optional[torch.Tensor] = None,
                 pieces_f16=None):
        assert x.dim() == 4 and x.shape[1] == 3 and bank.shape[1:] == (3, 5, 5)
        self.x = x.contiguous(memory_format=torch.channels_last)
        self.bank = bank.contiguous(memory_format=torch.channels_last)           # memory order (C, 5, 5, 3)
        self.bias, self.relu = (bias.contiguous() if bias is not None else None), bool(relu)
        self.shape = torch.Size((x.shape[0], bank.shape[0], x.shape[2] - 4, x.shape[3] - 4))
        self.device, self.dtype, self.is_cuda = x.device, x.dtype, x.is_cuda
        self._pieces = pieces
        self._pieces_f16 = pieces_f16

    def pieces_f16(self):
        """(wh, w_scale): the folded bank as the operand of eqa_lift5_fft48k5_input_f16x2 -- (C, 2 pieces, 5 filter rows, 4 chunks, 8)
        fp16 of w_scale * w (w_scale: the power of two that takes max |w| to at most 2^14); chunk p < 3 of filter row ky =
        [w(ci 0..2, kx = 2 p - 1), 0, w(ci 0..2, kx = 2 p), 0] (kx = -1: 0), chunk 3 = 0; pieces h1 = rn16(v), h2 = rn16(v - h1).
        Built once per LiftedInput (one host synchronisation); the network caches it per weight version."""
        if self._pieces_f16 is None:
            w = self.bank.float()                                   # (C, 3, 5, 5) logical
            C = w.shape[0]
            mx = float(w.abs().max().item())
            ex = 14 - math.frexp(mx)[1] if 0.0 < mx < float("inf") else 0
            scale = math.ldexp(1.0, max(-100, min(100, ex)))
            wp = torch.zeros(C, 5, 8, 4, dtype=torch.float32, device=w.device)       # (co, ky, kx + 1 in 0..7, ci padded to 4)
            wp[:, :, 1:6, :3] = w.permute(0, 2, 3, 1) * scale
            chunks = wp.reshape(C, 5, 4, 8)
            h1 = chunks.half()
            h2 = (chunks - h1.float()).half()
            self._pieces_f16 = (torch.stack([h1, h2], dim=1).contiguous(), scale)
        return self._pieces_f16

    def pieces(self) -> torch.Tensor:
        """The folded bank as the operand of eqa_lift5_fft48k5_input_bf16x3: (C, 3 pieces, 16 chunks, 8) bf16 -- chunk c < 15 = (filter
        row c // 3, pixel pair c % 3): [w(ci 0..2, kx = 2 p), 0, w(ci 0..2, kx = 2 p + 1), 0] (kx = 5: 0), chunk 15 = 0; every value
        split exactly into three bf16 pieces (built once per LiftedInput; the network caches it per weight version)."""
        if self._pieces is None:
            w = self.bank.float()                                   # (C, 3, 5, 5) logical
            C = w.shape[0]
            wp = torch.zeros(C, 5, 6, 4, dtype=torch.float32, device=w.device)       # (co, ky, kx padded to 6, ci padded to 4)
            wp[:, :, :5, :3] = w.permute(0, 2, 3, 1)
            chunks = torch.zeros(C, 16, 8, dtype=torch.float32, device=w.device)
            chunks[:, :15] = wp.reshape(C, 5, 3, 8).reshape(C, 15, 8)
            p0 = chunks.bfloat16()
            r1 = chunks - p0.float()
            p1 = r1.bfloat16()
            p2 = (r1 - p1.float()).bfloat16()
            assert torch.equal((p0.float() + p1.float()) + p2.float(), chunks), "the three bf16 pieces of a weight must add up exactly"
            self._pieces = torch.stack([p0, p1, p2], dim=1).contiguous()
        return self._pieces

    def materialize(self) -> torch.Tensor:
        """The map itself, channels-last (the unfused lifting kernel) -- for a consumer that turned out not to be `conv5x5`."""
        y = ops.lift_conv_nhwc(self.x, ops.pack_lift_weights(self.bank), self.bias, self.relu, 5, 5)
        return y


# how the fused kernel multiplies: "h2" (default since the end of round 6) = two fp16 pieces per value, three exact products on the fp16
# matrix cores (eqa_lift5_fft48k5_input_f16x2; pixels scaled under the bound eqa_absmax_slots takes, one 10 us launch; as close to
# fp64 as the fp32 form: tests/test_gpu_lift_fft.py), "f32" = v_mfma_f32_16x16x4_f32 (which shares the vector ALU's datapath with the
# transforms), "bf16x3" = exact three-piece splits, six products on the bf16 matrix cores (opt-in: no faster than f32)
LIFT_FFT_FORM = os.environ.get("EQA_LIFT_FFT_FORM", "h2")
LIFT_FFT_FUSED_DEFAULT = "1"     # EQA_LIFT_FFT_FUSED=0: the two kernels (eqa_lift_conv_grouped, eqa_fft48k5_input_grouped)


def lift_fused_applicable(x_shape, bank_shape, cout_next: int, device) -> bool:
    """Can the lifting layer (x_shape channels-last input, bank_shape (C, Cin, k, k)) be fused into the FFT convolution behind it?"""
    if os.environ.get("EQA_LIFT_FFT_FUSED", LIFT_FFT_FUSED_DEFAULT) == "0" or len(x_shape) != 4:
        return False
    C, Cin, kh, kw = bank_shape
    if not _lib.load().eqa_lift5_fft48k5_input_supported(Cin, kh, kw, C):
        return False
    out_shape = (x_shape[0], C, x_shape[2] - 4, x_shape[3] - 4)
    return x_shape[2] >= 5 and x_shape[3] >= 5 and grouped_applicable(out_shape, C, cout_next, device)


def grouped_applicable(shape, cin: int, cout: int, device, max_waste: float = 1.12) -> bool:
    """Would `conv5x5` take a (nimg, cin, H, W) fp32 map of this shape on `device` through the FFT path AND read it in the
    grouped layout?  (Decided before the producing layer runs, so that it can write that layout.)"""
    if not (ENABLED and device.type == "cuda" and len(shape) == 4 and shape[1] == cin and cin % 16 == 0):
        return False
    if os.environ.get("EQA_FFT_GROUPED", "1") == "0" or not _lib.load().eqa_fft48k5_input_grouped_supported(cin):
        return False
    if not gemm3m_supported(cin, cout) and ops.plane_gemm_supported(cin, cout):   # see `applicable`
        return False
    H, W = shape[-2:]
    if H < 16 or W < 16 or shape[0] * tiles(H) * tiles(W) < MIN_TILES:
        return False
    return tiles(H) * OUT <= max_waste * (H - 4) and tiles(W) * OUT <= max_waste * (W - 4)


def applicable(x: torch.Tensor, cin: int, cout: int, max_waste: float = 1.12) -> bool:
    """Channels-last fp32 device tensor, 5x5 kernel, enough tiles to amortise the filter spectra (``MIN_TILES``), and tiles
    that fit the output to within ``max_waste`` (the FFT work is per tile: 88 outputs per axis = 2 tiles exactly, 84 would
    waste 5 %, 50 would waste 43 % -> Winograd)."""
    if isinstance(x, (GroupedMap, LiftedInput)):
        return grouped_applicable(x.shape, cin, cout, x.device, max_waste)
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        return False
    H, W = x.shape[-2:]
    if H < 16 or W < 16 or x.shape[1] != cin:
        return False
    if not gemm3m_supported(cin, cout) and ops.plane_gemm_supported(cin, cout):
        # (e.g. 32 output channels: the complex GEMM wants 64-column tiles, the Winograd planes' GEMM takes 32 -- the path whose
        # contraction is hand-written wins over the one that would call the library)
        return False
    oh, ow = H - 4, W - 4
    if x.shape[0] * tiles(H) * tiles(W) < MIN_TILES:
        return False
    return tiles(H) * OUT <= max_waste * oh and tiles(W) * OUT <= max_waste * ow


def _order(C: int, G: int) -> torch.Tensor:
    """Position p of a 2C-wide row -> index into [Re(0..C-1), Im(0..C-1)]: groups of G channels, [Re x G | Im x G] each."""
    p = torch.arange(2 * C)
    grp, r = p // (2 * G), p % (2 * G)
    return (r // G) * C + grp * G + r % G


def group_sizes(cin: int, cout: int):
    """Channels per [Re | Im] group in the rows of V (16: what a block of the fused forward kernel owns) and of Mo (1)."""
    return (16 if cin % 16 == 0 else 1), 1


def filter_spectra(bank: torch.Tensor, groups=None, correlate: bool = True) -> torch.Tensor:
    """(Cout, Cin, 5, 5) -> B:(F, 2 Cin, 2 Cout) fp32, the real form of conj(FFT(filter)) / 48^2 per frequency.  Rows
    follow the rows of V, columns the rows of Mo (``group_sizes``; ``groups`` = (Cin, Cout) gives the plain [Re | Im] order)."""
    Cout, Cin = bank.shape[:2]
    if groups is None and bank.is_cuda and bank.dtype == torch.float32:
        lib = _lib.load()                                         # on the device: one kernel (eqa_fft48k5_filter_spectra)
        B = torch.empty((F, 2 * Cin, 2 * Cout), dtype=torch.float32, device=bank.device)
        with torch.cuda.device(bank.device):
            _lib.check(lib.eqa_fft48k5_filter_spectra(bank.contiguous().data_ptr(), B.data_ptr(), Cout, Cin, int(correlate),
                                                      torch.cuda.current_stream().cuda_stream), "eqa_fft48k5_filter_spectra")
        return B
    assert correlate, "the host construction is the correlation (forward) form"
    gin, gout = groups if groups is not None else group_sizes(Cin, Cout)
    wp = torch.zeros(Cout, Cin, N, N, dtype=torch.float64, device=bank.device)
    wp[:, :, :5, :5] = bank.double()
    W = torch.fft.rfft2(wp).conj() / float(N * N)                # (Cout, Cin, 48, 25)
    ky, kx = freq_index()
    W = W[:, :, ky.to(W.device), kx.to(W.device)]                # (Cout, Cin, F)
    Wr = W.real.permute(2, 1, 0).contiguous()
    Wi = W.imag.permute(2, 1, 0).contiguous()
    top = torch.cat([Wr, Wi], dim=2)                              # rows Re(A): [ Br |  Bi ]
    bot = torch.cat([-Wi, Wr], dim=2)                             # rows Im(A): [-Bi |  Br ]
    full = torch.cat([top, bot], dim=1)                           # rows [Re ci | Im ci], columns [Re co | Im co]
    full = full[:, _order(Cin, gin).to(bank.device)][:, :, _order(Cout, gout).to(bank.device)]
    return full.float().contiguous()


def spectra_buffer(M: int, width: int, device) -> torch.Tensor:
    """(F, M, width) fp32 view of a buffer with eqa_fft48k5_tile_pitch(M) = M | 1 rows per frequency -- the layout every
    eqa_fft48k5_* entry point reads and writes (an even tile count such as 1024 would put all the frequencies of a tile
    into the same HBM channel).  The batched GEMMs run on the view; the padding row is never touched."""
    pitch = _lib.load().eqa_fft48k5_tile_pitch(M)
    return torch.empty((F, pitch, width), dtype=torch.float32, device=device)[:, :M]


def output_stats_supported(nimg: int, OH: int, OW: int, cout: int) -> bool:
    """The inverse transform can take the batch-norm statistics of its output on the way out (eqa_fft48k5_output_stats)."""
    return _lib.load().eqa_fft48k5_output_stats_rows(nimg, OH, OW, cout) > 0


def conv5x5(x: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor], relu: bool,
            in_bias: Optional[torch.Tensor] = None, in_relu: bool = False, sums_k: int = 0,
            keep_V: Optional[list] = None, stats: Optional[list] = None) -> torch.Tensor:
    """x: channels-last (nimg,Cin,H,W) -> channels-last (nimg,Cout,H-4,W-4) = [relu](conv2d(act(x), g) + bias) with
    B = spectra_for(g) (either form) and act(x) = [relu](x + in_bias[c]) applied while loading; ``sums_k`` > 0: return instead the
    (nimg, Cout, sums_k, sums_k) fp64 window sums of that output (the linearised last layer consumes only those).
    ``keep_V``: a list that receives the input spectra V (training: the filter gradient reuses them).  ``stats``: a list that
    receives the (rows, Cout, 2) fp64 partial sums of the output's per-channel sum / sum of squares, taken by the inverse
    transform (training, no bias / activation; the caller checks ``output_stats_supported``)."""
    lib = _lib.load()
    nimg, Cin, H, W = x.shape
    if isinstance(B, Spectra3M):
        Cout = B.cout
        assert B.cin == Cin
    else:
        Cout = B.shape[2] // 2
        assert B.shape == (F, 2 * Cin, 2 * Cout)
    OH, OW = H - 4, W - 4
    TY, TX = tiles(H), tiles(W)
    M = nimg * TY * TX
    dev = x.device
    st = torch.cuda.current_stream().cuda_stream
    V = spectra_buffer(M, 2 * Cin, dev)
    p_in_bias = in_bias.data_ptr() if in_bias is not None else None
    p_bias = bias.data_ptr() if bias is not None else None
    vbound = None
    with torch.cuda.device(dev):
        if isinstance(x, LiftedInput):
            assert in_bias is None and not in_relu
            with _timed("lift_fft_input"):
                p_b = x.bias.data_ptr() if x.bias is not None else None
                if LIFT_FFT_FORM == "h2" and x.x.data_ptr() % 16 == 0:      # (eqa_absmax_slots reads 16 bytes per lane)
                    wh, w_scale = x.pieces_f16()
                    xbound = torch.empty(DCMAX_SLOTS, dtype=torch.float32, device=dev)
                    _lib.check(lib.eqa_absmax_slots(x.x.data_ptr(), x.x.numel(), xbound.data_ptr(), st), "eqa_absmax_slots")
                    want_dc = x.relu and isinstance(B, Spectra3M) and gemm_form(Cin, Cout, True) == "h3"
                    if want_dc:
                        vbound = torch.empty(DCMAX_SLOTS, dtype=torch.float32, device=dev)
                    _lib.check(lib.eqa_lift5_fft48k5_input_f16x2(x.x.data_ptr(), wh.data_ptr(), w_scale, xbound.data_ptr(), DCMAX_SLOTS, p_b,
                                                                 int(x.relu), V.data_ptr(), vbound.data_ptr() if want_dc else None, nimg,
                                                                 H + 4, W + 4, Cin, st), "eqa_lift5_fft48k5_input_f16x2")
                elif LIFT_FFT_FORM == "bf16x3":
                    _lib.check(lib.eqa_lift5_fft48k5_input_bf16x3(x.x.data_ptr(), x.pieces().data_ptr(), p_b, int(x.relu), V.data_ptr(), nimg,
                                                                  H + 4, W + 4, Cin, st), "eqa_lift5_fft48k5_input_bf16x3")
                elif x.relu and isinstance(B, Spectra3M) and gemm_form(Cin, Cout, True) == "h3":
                    # non-negative activations: the kernel also hands over its DC bins, the bound the fp16 contraction scales by
                    vbound = torch.empty(DCMAX_SLOTS, dtype=torch.float32, device=dev)
                    _lib.check(lib.eqa_lift5_fft48k5_input_dcmax(x.x.data_ptr(), x.bank.data_ptr(), p_b, 1, V.data_ptr(), vbound.data_ptr(),
                                                                 nimg, H + 4, W + 4, Cin, st), "eqa_lift5_fft48k5_input_dcmax")
                else:
                    _lib.check(lib.eqa_lift5_fft48k5_input(x.x.data_ptr(), x.bank.data_ptr(), p_b, int(x.relu), V.data_ptr(), nimg, H + 4, W + 4,
                                                           Cin, st), "eqa_lift5_fft48k5_input")
        else:
            T = torch.empty(max(lib.eqa_fft48k5_workspace_bytes(nimg, H, OW, Cin), 4) // 4, dtype=torch.float32, device=dev)
            with _timed("fft_input"):
                fn = lib.eqa_fft48k5_input_grouped if isinstance(x, GroupedMap) else lib.eqa_fft48k5_input
                _lib.check(fn(x.data_ptr(), T.data_ptr(), V.data_ptr(), p_in_bias, int(in_relu), nimg, H, W, Cin, st), "eqa_fft48k5_input")
            del T
        with _timed("fft_gemm"):
            Mo = contract(V, B, M, vbound)
        if keep_V is not None:
            keep_V.append(V)
        del V
        T2 = torch.empty(max(lib.eqa_fft48k5_workspace_bytes(nimg, OH, OW, Cout), 4) // 4, dtype=torch.float32, device=dev)
        if sums_k:
            S = torch.empty((nimg, Cout, sums_k, sums_k), dtype=torch.float64, device=dev)
            ws = torch.empty(nimg * OH * TX * Cout * (2 * sums_k - 1), dtype=torch.float32, device=dev)
            with _timed("fft_output_sums"):
                _lib.check(lib.eqa_fft48k5_output_sums(Mo.data_ptr(), T2.data_ptr(), p_bias, int(relu), S.data_ptr(), ws.data_ptr(),
                                                       nimg, OH, OW, Cout, sums_k, st), "eqa_fft48k5_output_sums")
            return S
        y = torch.empty((nimg, Cout, OH, OW), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        if stats is not None:
            assert bias is None and not relu
            part = torch.empty((lib.eqa_fft48k5_output_stats_rows(nimg, OH, OW, Cout), Cout, 2), dtype=torch.float64, device=dev)
            with _timed("fft_output"):
                _lib.check(lib.eqa_fft48k5_output_stats(Mo.data_ptr(), T2.data_ptr(), y.data_ptr(), part.data_ptr(), nimg, OH, OW, Cout, st),
                           "eqa_fft48k5_output_stats")
            stats.append(part)
            return y
        with _timed("fft_output"):
            _lib.check(lib.eqa_fft48k5_output(Mo.data_ptr(), T2.data_ptr(), p_bias, int(relu), y.data_ptr(), nimg, OH, OW, Cout, st),
                       "eqa_fft48k5_output")
        return y


def filter_grad(V: torch.Tensor, dy: torch.Tensor, cin: int, G: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d loss / d filter bank (Cout, Cin, 5, 5) of y = conv2d(x, bank) from V = the spectra of x's tiles (``keep_V`` of the
    forward pass) and the output gradient dy (channels-last): spectra of the disjoint 44 x 44 gradient tiles, one batched GEMM
    over the tiles per frequency (D[f] = V[f]^T G[f]), and the inverse transform restricted to the 5 x 5 support
    (eqa_fft48k5_grad_transform / _filter_grad).  2.5 multiplies per output as in the forward pass; Winograd's filter gradient
    (`winograd.filter_grad`) needs 4 and a 4x larger transformed gradient."""
    lib = _lib.load()
    nimg, Cout, OH, OW = dy.shape
    dev = dy.device
    M = V.shape[1]
    assert V.shape == (F, M, 2 * cin) and M == nimg * tiles(OH + 4) * tiles(OW + 4)
    st = torch.cuda.current_stream().cuda_stream
    if G is None:
        G = grad_spectra(dy)
    dbank = torch.empty((Cout, cin, 5, 5), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if GEMM == "3m" and lib.eqa_fft48k5_wgrad3m_supported(cin, Cout) and os.environ.get("EQA_FFT_WGRAD3M", "1") != "0":
            # the contraction over the tiles in the 3-multiplication form on the fp32 MFMA (the library's real GEMM: 4 products)
            D = torch.empty((F, cin, 2, Cout), dtype=torch.float32, device=dev)      # Dr | Di per input channel
            _lib.check(lib.eqa_fft48k5_wgrad3m(V.data_ptr(), G.data_ptr(), D.data_ptr(), M, cin, Cout, st), "eqa_fft48k5_wgrad3m")
            _lib.check(lib.eqa_fft48k5_filter_grad3m(D.data_ptr(), dbank.data_ptr(), Cout, cin, st), "eqa_fft48k5_filter_grad3m")
            return dbank
        D = torch.bmm(V.transpose(1, 2), G)                       # (F, 2 Cin, 2 Cout)
        _lib.check(lib.eqa_fft48k5_filter_grad(D.data_ptr(), dbank.data_ptr(), Cout, cin, st), "eqa_fft48k5_filter_grad")
    return dbank


def input_grad(dy: torch.Tensor, bank: torch.Tensor, G: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d loss / d x (channels-last (nimg, Cin, OH+4, OW+4)) of y = conv2d(x, bank): the full convolution of dy with the
    filters, tile by tile in the frequency domain with overlap-add (eqa_fft48k5_input_grad).  ``G``: the gradient-tile
    spectra if the caller already has them (`grad_spectra`)."""
    lib = _lib.load()
    nimg, Cout, OH, OW = dy.shape
    Cin = bank.shape[1]
    dev = dy.device
    if G is None:
        G = grad_spectra(dy)
    B2 = spectra_for(bank.detach().permute(1, 0, 2, 3).contiguous(), correlate=False)         # spectra of (Cin, Cout, 5, 5)
    st = torch.cuda.current_stream().cuda_stream
    H, W = OH + 4, OW + 4
    dx = torch.empty((nimg, Cin, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    with torch.cuda.device(dev):
        Cg = contract(G, B2, G.shape[1])                                          # (F, M, 2 Cin)
        T2 = torch.empty(max(lib.eqa_fft48k5_workspace_bytes(nimg, N * tiles(H), OW, Cin), 4) // 4, dtype=torch.float32, device=dev)
        _lib.check(lib.eqa_fft48k5_input_grad(Cg.data_ptr(), T2.data_ptr(), dx.data_ptr(), nimg, H, W, Cin, st), "eqa_fft48k5_input_grad")
    return dx


def grad_spectra(dy: torch.Tensor) -> torch.Tensor:
    """Spectra of the disjoint 44 x 44 tiles of an output gradient (channels-last), (F, M, 2 Cout): shared by the filter
    gradient and the input gradient."""
    lib = _lib.load()
    nimg, Cout, OH, OW = dy.shape
    M = nimg * tiles(OH + 4) * tiles(OW + 4)
    T = torch.empty(max(lib.eqa_fft48k5_workspace_bytes(nimg, OH, OW, Cout), 4) // 4, dtype=torch.float32, device=dy.device)
    G = spectra_buffer(M, 2 * Cout, dy.device)
    with torch.cuda.device(dy.device):
        _lib.check(lib.eqa_fft48k5_grad_transform(dy.data_ptr(), T.data_ptr(), G.data_ptr(), nimg, OH, OW, Cout,
                                                  torch.cuda.current_stream().cuda_stream), "eqa_fft48k5_grad_transform")
    return G


# ----------------------------------------------------------------------------------------------------------------------------------
# Kernel sizes other than 5 (round 4): the same scheme with O = 49 - k outputs per tile, through eqa_fft48_* (two-pass kernels).
# The reference's kernel_size is a free constructor argument (escnn_networks.py:19-44): its tutorial trains k = 9, its test k = 3.
# k = 5 keeps the functions above (fused / pipelined transforms).
# ----------------------------------------------------------------------------------------------------------------------------------
KSIZES = (3, 5, 7, 9)


def tiles_k(n: int, k: int) -> int:
    o = N + 1 - k
    return 0 if n < k else (n - (k - 1) + o - 1) // o


def applicable_k(x_shape, cin: int, cout: int, k: int, device, channels_last: bool = True) -> bool:
    """Should a (nimg, cin, H, W) fp32 channels-last map on `device` take the FFT path for a k x k stride-1 convolution?
    k = 5: the tuned rule above (`applicable`).  Otherwise a cost model -- the FFT path moves ~2.2 x its spectra (two-pass
    transforms) at ~4 TB/s and contracts 1154 x 3 real products per tile and channel pair at ~100 TFLOP/s; the library's
    direct convolution runs its OH x OW x k^2 products at ~110 TFLOP/s -- with a 1.3 x margin in favour of the library."""
    if not (ENABLED and k in KSIZES and k != 5 and device.type == "cuda" and channels_last and len(x_shape) == 4 and x_shape[1] == cin):
        return False
    if os.environ.get("EQA_FFT_ANYK", "1") == "0":
        return False
    nimg, _, H, W = x_shape
    if H < max(16, k) or W < max(16, k) or nimg * tiles_k(H, k) * tiles_k(W, k) < 8:
        return False
    oh, ow = H - k + 1, W - k + 1
    t = tiles_k(H, k) * tiles_k(W, k)
    fft_us = t * (F * 8 * (cin + cout) * 2.2 / 4.0e6 + F * 6 * cin * cout / 100e6)
    direct_us = oh * ow * k * k * 2 * cin * cout / 110e6
    return fft_us * 1.3 < direct_us


def spectra_for_k(bank: torch.Tensor, correlate: bool = True):
    """`spectra_for` for a (Cout, Cin, k, k) device bank of any supported k."""
    k = bank.shape[-1]
    if k == 5:
        return spectra_for(bank, correlate)
    lib = _lib.load()
    Cout, Cin = bank.shape[:2]
    assert bank.is_cuda and bank.dtype == torch.float32 and bank.shape[-2] == k and lib.eqa_fft48_supported(k)
    st = torch.cuda.current_stream().cuda_stream
    with torch.cuda.device(bank.device):
        if gemm3m_supported(Cin, Cout):
            B3 = torch.empty(lib.eqa_fft48k5_spectra3m_floats(Cin, Cout), dtype=torch.float32, device=bank.device)
            _lib.check(lib.eqa_fft48_filter_spectra3m(bank.contiguous().data_ptr(), B3.data_ptr(), Cout, Cin, k, int(correlate), st),
                       "eqa_fft48_filter_spectra3m")
            return Spectra3M(B3, Cin, Cout)
        B = torch.empty((F, 2 * Cin, 2 * Cout), dtype=torch.float32, device=bank.device)
        _lib.check(lib.eqa_fft48_filter_spectra(bank.contiguous().data_ptr(), B.data_ptr(), Cout, Cin, k, int(correlate), st),
                   "eqa_fft48_filter_spectra")
        return B


def conv_kxk(x: torch.Tensor, B, k: int, bias: Optional[torch.Tensor], relu: bool, in_bias: Optional[torch.Tensor] = None,
             in_relu: bool = False, keep_V: Optional[list] = None) -> torch.Tensor:
    """`conv5x5` for any supported k (no window-sum / statistics epilogues off k = 5): channels-last (nimg, Cin, H, W) ->
    channels-last (nimg, Cout, H-k+1, W-k+1) = [relu](conv2d(act(x), g) + bias), B = spectra_for_k(g)."""
    if k == 5:
        return conv5x5(x, B, bias, relu, in_bias, in_relu, keep_V=keep_V)
    lib = _lib.load()
    nimg, Cin, H, W = x.shape
    Cout = B.cout if isinstance(B, Spectra3M) else B.shape[2] // 2
    OH, OW = H - k + 1, W - k + 1
    M = nimg * tiles_k(H, k) * tiles_k(W, k)
    dev = x.device
    st = torch.cuda.current_stream().cuda_stream
    T = torch.empty(max(lib.eqa_fft48_workspace_bytes(nimg, H, OW, Cin, k), 4) // 4, dtype=torch.float32, device=dev)
    V = spectra_buffer(M, 2 * Cin, dev)
    with torch.cuda.device(dev):
        with _timed("fft_input"):
            _lib.check(lib.eqa_fft48_input(x.data_ptr(), T.data_ptr(), V.data_ptr(), in_bias.data_ptr() if in_bias is not None else None,
                                           int(in_relu), nimg, H, W, Cin, k, st), "eqa_fft48_input")
        del T
        with _timed("fft_gemm"):
            Mo = contract(V, B, M)
        if keep_V is not None:
            keep_V.append(V)
        del V
        T2 = torch.empty(max(lib.eqa_fft48_workspace_bytes(nimg, OH, OW, Cout, k), 4) // 4, dtype=torch.float32, device=dev)
        y = torch.empty((nimg, Cout, OH, OW), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        with _timed("fft_output"):
            _lib.check(lib.eqa_fft48_output(Mo.data_ptr(), T2.data_ptr(), bias.data_ptr() if bias is not None else None, int(relu),
                                            y.data_ptr(), nimg, OH, OW, Cout, k, st), "eqa_fft48_output")
    return y


def grad_spectra_k(dy: torch.Tensor, k: int) -> torch.Tensor:
    if k == 5:
        return grad_spectra(dy)
    lib = _lib.load()
    nimg, Cout, OH, OW = dy.shape
    M = nimg * tiles_k(OH + k - 1, k) * tiles_k(OW + k - 1, k)
    T = torch.empty(max(lib.eqa_fft48_workspace_bytes(nimg, OH, OW, Cout, k), 4) // 4, dtype=torch.float32, device=dy.device)
    G = spectra_buffer(M, 2 * Cout, dy.device)
    with torch.cuda.device(dy.device):
        _lib.check(lib.eqa_fft48_grad_transform(dy.data_ptr(), T.data_ptr(), G.data_ptr(), nimg, OH, OW, Cout, k,
                                                torch.cuda.current_stream().cuda_stream), "eqa_fft48_grad_transform")
    return G


def filter_grad_k(V: torch.Tensor, dy: torch.Tensor, cin: int, k: int, G: Optional[torch.Tensor] = None) -> torch.Tensor:
    if k == 5:
        return filter_grad(V, dy, cin, G)
    lib = _lib.load()
    nimg, Cout, OH, OW = dy.shape
    dev = dy.device
    M = V.shape[1]
    assert V.shape == (F, M, 2 * cin) and M == nimg * tiles_k(OH + k - 1, k) * tiles_k(OW + k - 1, k)
    st = torch.cuda.current_stream().cuda_stream
    if G is None:
        G = grad_spectra_k(dy, k)
    dbank = torch.empty((Cout, cin, k, k), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if GEMM == "3m" and lib.eqa_fft48k5_wgrad3m_supported(cin, Cout) and os.environ.get("EQA_FFT_WGRAD3M", "1") != "0":
            D = torch.empty((F, cin, 2, Cout), dtype=torch.float32, device=dev)
            _lib.check(lib.eqa_fft48k5_wgrad3m(V.data_ptr(), G.data_ptr(), D.data_ptr(), M, cin, Cout, st), "eqa_fft48k5_wgrad3m")
            _lib.check(lib.eqa_fft48_filter_grad(D.data_ptr(), dbank.data_ptr(), Cout, cin, k, 1, st), "eqa_fft48_filter_grad")
            return dbank
        D = torch.bmm(V.transpose(1, 2), G)
        _lib.check(lib.eqa_fft48_filter_grad(D.data_ptr(), dbank.data_ptr(), Cout, cin, k, 0, st), "eqa_fft48_filter_grad")
    return dbank


def input_grad_k(dy: torch.Tensor, bank: torch.Tensor, G: Optional[torch.Tensor] = None) -> torch.Tensor:
    k = bank.shape[-1]
    if k == 5:
        return input_grad(dy, bank, G)
    lib = _lib.load()
    nimg, Cout, OH, OW = dy.shape
    Cin = bank.shape[1]
    dev = dy.device
    if G is None:
        G = grad_spectra_k(dy, k)
    B2 = spectra_for_k(bank.detach().permute(1, 0, 2, 3).contiguous(), correlate=False)
    st = torch.cuda.current_stream().cuda_stream
    H, W = OH + k - 1, OW + k - 1
    dx = torch.empty((nimg, Cin, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    with torch.cuda.device(dev):
        Cg = contract(G, B2, G.shape[1])
        T2 = torch.empty(max(lib.eqa_fft48_workspace_bytes(nimg, N * tiles_k(H, k), OW, Cin, k), 4) // 4, dtype=torch.float32, device=dev)
        _lib.check(lib.eqa_fft48_input_grad(Cg.data_ptr(), T2.data_ptr(), dx.data_ptr(), nimg, H, W, Cin, k, st), "eqa_fft48_input_grad")
    return dx


class ConvKxKFunction(torch.autograd.Function):
    """y = conv2d(x, bank) (k x k, stride 1, no padding, channels-last) as an FFT convolution with both gradients in the frequency
    domain (one transform of the output gradient's disjoint tiles serves both) -- `winograd.Conv5x5Function`'s FFT branch for
    the kernel sizes Winograd F(m, 5) does not cover."""

    @staticmethod
    def forward(ctx, x, bank):
        keep: list = []
        k = bank.shape[-1]
        y = conv_kxk(x, spectra_for_k(bank.detach()), k, None, False, keep_V=keep)
        ctx.save_for_backward(bank, *keep)
        return y

    @staticmethod
    def backward(ctx, dy):
        bank, V = ctx.saved_tensors
        k = bank.shape[-1]
        dy = dy.contiguous(memory_format=torch.channels_last)
        G = grad_spectra_k(dy, k)
        dx = input_grad_k(dy, bank, G) if ctx.needs_input_grad[0] else None
        dbank = filter_grad_k(V, dy, bank.shape[1], k, G).to(bank.dtype) if ctx.needs_input_grad[1] else None
        return dx, dbank
