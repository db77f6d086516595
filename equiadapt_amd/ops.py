"""Tensor-level entry points over the C ABI (include/eqa_hip.h).

Every function takes ROCm ("cuda") fp32 tensors, launches on torch's current HIP stream and returns
torch tensors; nothing here computes on the CPU.  torch is plumbing only: device memory + streams.
"""
from typing import Optional, Tuple

import torch

from equiadapt_amd import _lib


MAX_WINDOW_K = 10   # == kMaxWinK (csrc/eqa_common.hpp): the largest window the window-sum kernels take (the linearised last layer)


def _stream() -> int:
    """The raw handle of torch's current HIP stream on the current device.  (torch.cuda.current_stream() builds a Stream object
    through four Python frames: 4-9 us per call, twelve calls in a configs[4] step -- a fifth of its host time at B = 4.)"""
    return _raw_stream(_current_device())


try:                                   # private but stable since torch 1.x (inductor's own launcher uses the first)
    _raw_stream, _current_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:                 # a torch without them: the public, slower spelling
    def _raw_stream(_dev):
        return torch.cuda.current_stream().cuda_stream

    def _current_device():
        return 0


class KernelTimer:
    """HIP-event bracket around named launches, recorded on the stream the kernels run on.

    ``with ops.KernelTimer() as t: ...`` then ``t.summary()`` -> {name: (launches, mean_ms)} after a
    device synchronise.  Used by bench.py for the live per-kernel duration behind ``roofline.achieved``.
    """

    active: "Optional[KernelTimer]" = None

    def __init__(self):
        self.events = {}

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def summary(self):
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.events.items()}


class _timed:
    def __init__(self, name: str):
        self.t = KernelTimer.active
        self.name = name

    def __enter__(self):
        if self.t is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.t is not None:
            self.b.record()
            self.t.events.setdefault(self.name, []).append((self.a, self.b))


def _need(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: equiadapt_amd runs on an MI355X (ROCm) device only, no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype} (the hot path computes in fp32 like the reference)")
    return t if t.is_contiguous() else t.contiguous()


def _opt(t: Optional[torch.Tensor], name: str, dtype) -> Tuple[Optional[torch.Tensor], Optional[int]]:
    if t is None:
        return None, None
    t = _need(t, name, dtype)
    return t, t.data_ptr()


def group_action(
    src: torch.Tensor,
    gidx: Optional[torch.Tensor],
    theta: torch.Tensor,
    flags: Optional[torch.Tensor],
    chan_map: Optional[torch.Tensor],
    pad: int,
    out_hw: Tuple[int, int],
    top_left: Tuple[int, int],
    n_out: Optional[int] = None,
) -> torch.Tensor:
    """Generic discrete group action on image planes (eqa_group_action_fwd)."""
    lib = _lib.load()
    src = _need(src, "src")
    theta = _need(theta, "theta")
    B, C, H, W = src.shape
    E = theta.shape[0]
    gidx, p_gidx = _opt(gidx, "gidx", torch.int32)
    flags, p_flags = _opt(flags, "flags", torch.int32)
    chan_map, p_map = _opt(chan_map, "chan_map", torch.int32)
    G = chan_map.shape[1] if chan_map is not None else 1
    if n_out is None:
        n_out = B if gidx is not None else E * B
    OH, OW = out_hw
    dst = torch.empty((n_out, C, OH, OW), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        st = lib.eqa_group_action_fwd(src.data_ptr(), dst.data_ptr(), p_gidx, theta.data_ptr(), p_flags, p_map, E, G,
                                      n_out, B, C, H, W, pad, OH, OW, top_left[0], top_left[1], _stream())
    _lib.check(st, "eqa_group_action_fwd")
    return dst


def group_action_bwd(
    src: torch.Tensor,
    grad_out: torch.Tensor,
    gidx: Optional[torch.Tensor],
    theta: torch.Tensor,
    flags: Optional[torch.Tensor],
    chan_map: Optional[torch.Tensor],
    pad: int,
    top_left: Tuple[int, int],
    want_src: bool,
    want_angle: bool,
    want_theta: bool = False,
) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Backward of ``group_action`` (eqa_group_action_bwd / eqa_group_action_bwd_theta).

    Returns (dL/d src or None, dL/d angle or None); the angle gradient is per output image and w.r.t. the
    ``angle`` argument IN DEGREES of the ``rotate(img, angle)`` that ``theta`` encodes.  With ``want_theta`` the second
    value is instead dL/d theta, (n_out, 6), for a per-output affine matrix (continuous groups).
    """
    lib = _lib.load()
    src = _need(src, "src")
    grad_out = _need(grad_out, "grad_out")
    theta = _need(theta, "theta")
    B, C, H, W = src.shape
    n_out, C2, OH, OW = grad_out.shape
    assert C2 == C
    E = theta.shape[0]
    gidx, p_gidx = _opt(gidx, "gidx", torch.int32)
    flags, p_flags = _opt(flags, "flags", torch.int32)
    chan_map, p_map = _opt(chan_map, "chan_map", torch.int32)
    G = chan_map.shape[1] if chan_map is not None else 1
    if want_src and pad > 0 and gidx is not None and n_out == B and chan_map is None and not lib.eqa_get_option(0):
        # edge-padded transform (canonicalize): deterministic, atomics-free input gradient in two steps -- the gather adjoint
        # on the padded FRAME as its source (dL/d frame), then the adjoint of the replicate padding (strips and corners folded
        # onto the image borders).  The transform gradient, if wanted, comes from the ordinary call below without grad_src.
        Hp, Wp = H + 2 * pad, W + 2 * pad
        g_frame = torch.empty((B, C, Hp, Wp), dtype=torch.float32, device=src.device)
        with torch.cuda.device(src.device):
            st = lib.eqa_group_action_bwd(g_frame.data_ptr(), grad_out.data_ptr(), p_gidx, theta.data_ptr(), p_flags, None,
                                          g_frame.data_ptr(), None, E, 1, n_out, B, C, Hp, Wp, 0, OH, OW, top_left[0], top_left[1],
                                          _stream())
            _lib.check(st, "eqa_group_action_bwd (frame gather)")
            g_src = torch.empty_like(src)
            ws = torch.empty((lib.eqa_fold_edge_pad_workspace_bytes(B * C, H, W, pad) // 4,), dtype=torch.float32, device=src.device)
            _lib.check(lib.eqa_fold_edge_pad(g_frame.data_ptr(), g_src.data_ptr(), ws.data_ptr(), B * C, H, W, pad, _stream()),
                       "eqa_fold_edge_pad")
        if not (want_angle or want_theta):
            return g_src, None
        _, g_t = group_action_bwd(src, grad_out, gidx, theta, flags, chan_map, pad, top_left, False, want_angle, want_theta)
        return g_src, g_t
    g_src = torch.zeros_like(src) if want_src else None
    tiles = lib.eqa_group_action_bwd_tiles(OH, OW)
    if want_theta:
        partial = torch.empty((n_out, tiles, 6), dtype=torch.float32, device=src.device)
        with torch.cuda.device(src.device):
            st = lib.eqa_group_action_bwd_theta(src.data_ptr(), grad_out.data_ptr(), p_gidx, theta.data_ptr(), p_flags, p_map,
                                                g_src.data_ptr() if want_src else None, partial.data_ptr(),
                                                E, G, n_out, B, C, H, W, pad, OH, OW, top_left[0], top_left[1], _stream())
        _lib.check(st, "eqa_group_action_bwd_theta")
        return g_src, partial.sum(dim=1)
    partial = torch.empty((n_out, tiles), dtype=torch.float32, device=src.device) if want_angle else None
    with torch.cuda.device(src.device):
        st = lib.eqa_group_action_bwd(src.data_ptr(), grad_out.data_ptr(), p_gidx, theta.data_ptr(), p_flags, p_map,
                                      g_src.data_ptr() if want_src else None, partial.data_ptr() if want_angle else None,
                                      E, G, n_out, B, C, H, W, pad, OH, OW, top_left[0], top_left[1], _stream())
    _lib.check(st, "eqa_group_action_bwd")
    g_angle = partial.sum(dim=1) * (3.141592653589793 / 180.0) if want_angle else None
    return g_src, g_angle


# Window-bound hints: device address of an element table -> the most source rows a 32 x 32 output tile can sample under ANY of its
# elements (images.utils.device_tables registers 35 for right-angle groups on square frames).  A registry keyed by the table's
# storage, not an attribute on the tensor: .to() / .clone() / tracing drop attributes silently, and a copy at another address
# simply has no hint.  The hint only sizes the LDS reservation (eqa_group_action_fwd_hint); a tile whose window exceeds it takes
# the direct gather path, so a stale or wrong hint costs time, never correctness.
#  keyed by (device, address, element count) and dropped when the registered tensor dies: the caching allocator hands a
# freed table's address to the next allocation, and another table at that address must not inherit the hint (it would push every
# oversized tile onto the slow path without anyone noticing).
_window_hints: dict = {}


def _hint_key(theta: torch.Tensor):
    return (theta.device.index, theta.data_ptr(), theta.numel())


def register_window_hint(theta: torch.Tensor, max_window_rows: int) -> None:
    import weakref

    key = _hint_key(theta)
    _window_hints[key] = int(max_window_rows)
    weakref.finalize(theta, _window_hints.pop, key, None)


def _window_hint(theta: torch.Tensor, explicit: Optional[int]) -> int:
    return int(explicit) if explicit is not None else _window_hints.get(_hint_key(theta), 0)


def canon_transform(x: torch.Tensor, gidx: torch.Tensor, theta: torch.Tensor, flags: Optional[torch.Tensor], pad: int,
                    max_window: Optional[int] = None) -> torch.Tensor:
    """I5: fused pad(edge) -> [hflip] -> rotate -> center-crop (eqa_canon_transform_fwd).  ``max_window``: bound on a tile's source
    window rows (0 = none; default: what was registered for this table)."""
    lib = _lib.load()
    x = _need(x, "x")
    gidx = _need(gidx, "gidx", torch.int32)
    theta = _need(theta, "theta")
    flags, p_flags = _opt(flags, "flags", torch.int32)
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    hint = _window_hint(theta, max_window)
    with torch.cuda.device(x.device), _timed("canon_transform"):
        if hint > 0 and B > 0:
            st = lib.eqa_group_action_fwd_hint(x.data_ptr(), y.data_ptr(), gidx.data_ptr(), theta.data_ptr(), p_flags, None,
                                               theta.shape[0], 1, B, B, C, H, W, pad, H, W, pad, pad, hint, _stream())
        else:
            st = lib.eqa_canon_transform_fwd(x.data_ptr(), y.data_ptr(), gidx.data_ptr(), theta.data_ptr(), p_flags,
                                             theta.shape[0], B, C, H, W, pad, _stream())
    _lib.check(st, "eqa_canon_transform_fwd")
    return y


def invert_action(f: torch.Tensor, gidx: torch.Tensor, theta: torch.Tensor, flags: Optional[torch.Tensor],
                  chan_map: Optional[torch.Tensor], max_window: Optional[int] = None) -> torch.Tensor:
    """I7: rotate(+theta) zero-corner -> flip -> regular-representation roll (eqa_invert_action_fwd)."""
    lib = _lib.load()
    f = _need(f, "feature_map")
    gidx = _need(gidx, "gidx", torch.int32)
    theta = _need(theta, "theta")
    flags, p_flags = _opt(flags, "flags", torch.int32)
    chan_map, p_map = _opt(chan_map, "chan_map", torch.int32)
    G = chan_map.shape[1] if chan_map is not None else 1
    B, C, H, W = f.shape
    out = torch.empty_like(f)
    hint = _window_hint(theta, max_window)
    with torch.cuda.device(f.device), _timed("invert_action"):
        if hint > 0 and B > 0:
            st = lib.eqa_group_action_fwd_hint(f.data_ptr(), out.data_ptr(), gidx.data_ptr(), theta.data_ptr(), p_flags, p_map,
                                               theta.shape[0], G, B, B, C, H, W, 0, H, W, 0, 0, hint, _stream())
        else:
            st = lib.eqa_invert_action_fwd(f.data_ptr(), out.data_ptr(), gidx.data_ptr(), theta.data_ptr(), p_flags, p_map,
                                           theta.shape[0], G, B, C, H, W, _stream())
    _lib.check(st, "eqa_invert_action_fwd")
    return out


def group_action_pair(x: torch.Tensor, f: torch.Tensor, gidx: torch.Tensor, theta_canon: torch.Tensor,
                      flags_canon: Optional[torch.Tensor], pad: int, theta_inv: torch.Tensor, flags_inv: Optional[torch.Tensor],
                      chan_map: Optional[torch.Tensor]):
    """I5 + I7 in one launch (eqa_group_action_pair): (canon_transform(x, ...), invert_action(f, ...)) for the same group
    index, bit-identical to the two separate calls.  x:(B,C,H,W), f:(B,Cf,H,W)."""
    lib = _lib.load()
    x, f = _need(x, "x"), _need(f, "feature_map")
    gidx = _need(gidx, "gidx", torch.int32)
    theta_canon, theta_inv = _need(theta_canon, "theta_canon"), _need(theta_inv, "theta_inv")
    flags_canon, p_fc = _opt(flags_canon, "flags_canon", torch.int32)
    flags_inv, p_fi = _opt(flags_inv, "flags_inv", torch.int32)
    chan_map, p_map = _opt(chan_map, "chan_map", torch.int32)
    G = chan_map.shape[1] if chan_map is not None else 1
    B, C, H, W = x.shape
    if f.shape[0] != B or tuple(f.shape[2:]) != (H, W) or theta_canon.shape[0] != theta_inv.shape[0]:
        raise ValueError(f"group_action_pair: x {tuple(x.shape)} and f {tuple(f.shape)} must share batch and spatial size, and the tables their element count")
    y, out = torch.empty_like(x), torch.empty_like(f)
    with torch.cuda.device(x.device), _timed("group_action_pair"):
        st = lib.eqa_group_action_pair(x.data_ptr(), y.data_ptr(), theta_canon.data_ptr(), p_fc, pad, C, f.data_ptr(), out.data_ptr(),
                                       theta_inv.data_ptr(), p_fi, p_map, G, f.shape[1], gidx.data_ptr(), theta_canon.shape[0], B, H, W,
                                       _stream())
    _lib.check(st, "eqa_group_action_pair")
    return y, out


def orbit_expand(x: torch.Tensor, theta: torch.Tensor, flags: Optional[torch.Tensor], pad: int) -> torch.Tensor:
    """I8: all E group views of every image, element-major (eqa_orbit_expand_fwd)."""
    lib = _lib.load()
    x = _need(x, "x")
    theta = _need(theta, "theta")
    flags, p_flags = _opt(flags, "flags", torch.int32)
    B, C, S, S2 = x.shape
    if S != S2:
        raise ValueError("orbit_expand expects square images (the reference crops to a square resize_shape)")
    E = theta.shape[0]
    y = torch.empty((E * B, C, S, S), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = lib.eqa_orbit_expand_fwd(x.data_ptr(), y.data_ptr(), theta.data_ptr(), p_flags, E, B, C, S, pad, _stream())
    _lib.check(st, "eqa_orbit_expand_fwd")
    return y


def group_pool_argmax(feature_map: torch.Tensor, want_index: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """I3+I4: (B, Cf, G, H', W') -> activations (B, G) [mean over Cf, H', W'] and argmax index (B,) int32."""
    lib = _lib.load()
    feature_map = _need(feature_map, "feature_map")
    B, Cf, G, Hf, Wf = feature_map.shape
    HW = Hf * Wf
    act = torch.empty((B, G), dtype=torch.float32, device=feature_map.device)
    gidx = torch.empty((B,), dtype=torch.int32, device=feature_map.device) if want_index else None
    nbytes = lib.eqa_group_pool_workspace_bytes(B, Cf, G, HW)
    ws = torch.empty((max(nbytes, 8) // 8,), dtype=torch.float64, device=feature_map.device)
    with torch.cuda.device(feature_map.device), _timed("group_pool"):
        st = lib.eqa_group_pool_argmax(feature_map.data_ptr(), act.data_ptr(), gidx.data_ptr() if want_index else None,
                                       ws.data_ptr(), B, Cf, G, HW, _stream())
    _lib.check(st, "eqa_group_pool_argmax")
    return act, gidx


def window_sums(x: torch.Tensor, k: int, scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
                relu: bool = False) -> torch.Tensor:
    """(B,C,H,W) fp32 -> (B,C,k,k) fp64 sums of act(x) over the k*k shifted (H-k+1)x(W-k+1) windows
    (eqa_window_sums); act(t) = [relu](scale[c]*t + shift[c])."""
    lib = _lib.load()
    B, C, H, W = x.shape
    scale, p_scale = _opt(scale, "scale", torch.float32)
    shift, p_shift = _opt(shift, "shift", torch.float32)
    out = torch.empty((B, C, k, k), dtype=torch.float64, device=x.device)
    if (x.is_cuda and x.dtype == torch.float32 and C % 4 == 0 and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last) and H > 2 * (k - 1) and W > 2 * (k - 1)):
        # (B,H,W,C) in memory: the channels-last kernels read it as it is
        ws = torch.empty((max(lib.eqa_window_sums_nhwc_workspace_bytes(B, C, H, k), 4) // 4,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device), _timed("window_sums"):
            st = lib.eqa_window_sums_nhwc(x.data_ptr(), p_scale, p_shift, int(relu), out.data_ptr(), ws.data_ptr(), B, C, H, W, k, _stream())
        _lib.check(st, "eqa_window_sums_nhwc")
        return out
    x = _need(x, "x")
    with torch.cuda.device(x.device), _timed("window_sums"):
        st = lib.eqa_window_sums(x.data_ptr(), p_scale, p_shift, int(relu), out.data_ptr(), B, C, H, W, k, _stream())
    _lib.check(st, "eqa_window_sums")
    return out


def group_argmax(act: torch.Tensor) -> torch.Tensor:
    """I4: first-maximum argmax over the group axis, (B, G) -> (B,) int32."""
    lib = _lib.load()
    act = _need(act.detach(), "group_activations")
    B, G = act.shape
    gidx = torch.empty((B,), dtype=torch.int32, device=act.device)
    with torch.cuda.device(act.device):
        st = lib.eqa_group_argmax(act.data_ptr(), gidx.data_ptr(), B, G, _stream())
    _lib.check(st, "eqa_group_argmax")
    return gidx


def so3_rotate(x: torch.Tensor, R: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """P4: y[b] = R[b] @ x[b] (or R[b]^T @ x[b]);  x:(B,3,N), R:(B,3,3)."""
    lib = _lib.load()
    x = _need(x, "x")
    R = _need(R, "R")
    B, three, N = x.shape
    if three != 3 or tuple(R.shape) != (B, 3, 3):
        raise ValueError("so3_rotate expects x:(B,3,N) and R:(B,3,3)")
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        st = lib.eqa_so3_rotate(x.data_ptr(), R.data_ptr(), y.data_ptr(), B, N, int(transpose), _stream())
    _lib.check(st, "eqa_so3_rotate")
    return y


def gram_schmidt(v: torch.Tensor) -> torch.Tensor:
    """P3: batched classical Gram-Schmidt on the rows of (B,3,3)."""
    lib = _lib.load()
    v = _need(v, "vectors")
    if v.dim() != 3 or v.shape[1:] != (3, 3):
        raise ValueError("gram_schmidt expects (B,3,3)")
    out = torch.empty_like(v)
    with torch.cuda.device(v.device):
        st = lib.eqa_gram_schmidt(v.data_ptr(), out.data_ptr(), v.shape[0], _stream())
    _lib.check(st, "eqa_gram_schmidt")
    return out


def vnsmall_forward(x: torch.Tensor, params: torch.Tensor, k: int = 20, pooling: str = "mean") -> torch.Tensor:
    """P1+P2 fused: (B,3,N) point clouds + packed VNSmall parameters -> (B,3,3) equivariant vectors (eqa_vnsmall_fwd).
    ``pooling`` "mean" (1310 packed floats) or "max" (1751: + the pooling layer's 21 x 21 direction map)."""
    lib = _lib.load()
    x = _need(x, "point_cloud")
    params = _need(params, "params")
    B, three, N = x.shape
    if pooling not in ("mean", "max"):
        raise ValueError(f"Pooling type {pooling} not supported")
    if three != 3 or params.numel() != (1310 if pooling == "mean" else 1751):
        raise ValueError("vnsmall_forward expects x:(B,3,N) and 1310 (mean) / 1751 (max) packed parameters")
    out = torch.empty((B, 3, 3), dtype=torch.float32, device=x.device)
    ws = torch.empty((max(lib.eqa_vnsmall_workspace_bytes(B, N), 4) // 4,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed("vnsmall_fwd"):
        st = lib.eqa_vnsmall_fwd(x.data_ptr(), params.data_ptr(), out.data_ptr(), ws.data_ptr(), B, N, k, int(pooling == "max"), _stream())
    _lib.check(st, "eqa_vnsmall_fwd")
    return out


def vnsmall_canonicalize(x: torch.Tensor, params: torch.Tensor, k: int = 20, pooling: str = "mean"):
    """P1-P4 in two launches (eqa_vnsmall_canonicalize): (B,3,N) clouds -> (network output vectors (B,3,3), their Gram-Schmidt
    frame R (B,3,3), canonical clouds R x (B,3,N))."""
    lib = _lib.load()
    x = _need(x, "point_cloud")
    params = _need(params, "params")
    B, three, N = x.shape
    if pooling not in ("mean", "max"):
        raise ValueError(f"Pooling type {pooling} not supported")
    if three != 3 or params.numel() != (1310 if pooling == "mean" else 1751):
        raise ValueError("vnsmall_canonicalize expects x:(B,3,N) and 1310 (mean) / 1751 (max) packed parameters")
    vec = torch.empty((B, 3, 3), dtype=torch.float32, device=x.device)
    R = torch.empty((B, 3, 3), dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    ws = torch.empty((max(lib.eqa_vnsmall_workspace_bytes(B, N), 4) // 4,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed("vnsmall_fwd"):
        st = lib.eqa_vnsmall_canonicalize(x.data_ptr(), params.data_ptr(), vec.data_ptr(), R.data_ptr(), y.data_ptr(), ws.data_ptr(),
                                          B, N, k, int(pooling == "max"), _stream())
    _lib.check(st, "eqa_vnsmall_canonicalize")
    return vec, R, y


def crop_resize_aa(x: torch.Tensor, tables, out_hw: Tuple[int, int]) -> torch.Tensor:
    """I1: centre crop + antialiased bilinear resize in one kernel (eqa_crop_resize_aa).  ``tables`` from
    ``geometry.aa_resize_tables`` already moved to x.device."""
    lib = _lib.load()
    x = _need(x, "x")
    wx, x0, wy, y0, K, max_rows, x_begin, x_span = tables
    B, C, H, W = x.shape
    y = torch.empty((B, C, out_hw[0], out_hw[1]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed("crop_resize_aa"):
        st = lib.eqa_crop_resize_aa(x.data_ptr(), y.data_ptr(), wx.data_ptr(), x0.data_ptr(), wy.data_ptr(), y0.data_ptr(),
                                    B * C, H, W, out_hw[0], out_hw[1], K, max_rows, x_begin, x_span, _stream())
    _lib.check(st, "eqa_crop_resize_aa")
    return y


def mask_action_nearest(masks: torch.Tensor, eidx: torch.Tensor, rtheta: torch.Tensor, flags: Optional[torch.Tensor]) -> torch.Tensor:
    """I6: nearest-neighbour rotation (+ optional pre-flip) of (n,H,W) uint8 masks, element index per mask."""
    lib = _lib.load()
    masks = _need(masks, "masks", torch.uint8)
    eidx = _need(eidx, "eidx", torch.int32)
    rtheta = _need(rtheta, "rtheta")
    flags, p_flags = _opt(flags, "flags", torch.int32)
    n, H, W = masks.shape
    out = torch.empty_like(masks)
    with torch.cuda.device(masks.device), _timed("mask_action"):
        st = lib.eqa_mask_action_nearest(masks.data_ptr(), out.data_ptr(), eidx.data_ptr(), rtheta.data_ptr(), p_flags,
                                         rtheta.shape[0], n, H, W, _stream())
    _lib.check(st, "eqa_mask_action_nearest")
    return out


_plane_tables: dict = {}


def mask_action_nearest_planes(mask_list, eidx: torch.Tensor, rtheta: torch.Tensor, flags: Optional[torch.Tensor]) -> torch.Tensor:
    """I6 for a LIST of per-sample (n_t, H, W) uint8 mask tensors: one launch over all planes, read through a table of plane
    pointers (eqa_mask_action_nearest_planes) instead of a concatenated copy.  Returns one (sum n_t, H, W) tensor."""
    lib = _lib.load()
    eidx = _need(eidx, "eidx", torch.int32)
    rtheta = _need(rtheta, "rtheta")
    flags, p_flags = _opt(flags, "flags", torch.int32)
    masks = [_need(m, "masks", torch.uint8) for m in mask_list]
    H, W = masks[0].shape[-2:]
    ptrs = tuple(m.data_ptr() + k * H * W for m in masks for k in range(m.shape[0]))
    n = len(ptrs)
    dev = masks[0].device
    # the pointer table is uploaded once per set of mask tensors: a loop that re-uses its target buffers (and a captured
    # hipGraph, which must not contain a pageable host-to-device copy) finds it on the device
    key = (ptrs, str(dev))
    table = _plane_tables.get(key)
    if table is None:
        if len(_plane_tables) >= 64:
            _plane_tables.clear()
        table = torch.tensor(ptrs, dtype=torch.int64).to(dev)
        _plane_tables[key] = table
    out = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), _timed("mask_action"):
        st = lib.eqa_mask_action_nearest_planes(table.data_ptr(), out.data_ptr(), eidx.data_ptr(), rtheta.data_ptr(), p_flags,
                                                rtheta.shape[0], n, H, W, _stream())
    _lib.check(st, "eqa_mask_action_nearest_planes")
    return out


def boxes_action(boxes: torch.Tensor, img_of_box: torch.Tensor, rotation_deg: torch.Tensor, width: float,
                 flip_all: bool) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """I6: every (n,4) xyxy box flipped (if ``flip_all``) and rotated by its image's angle about (width/2, width/2), one
    launch (eqa_boxes_action).  Returns (new boxes, boxes after the flip or None)."""
    lib = _lib.load()
    boxes = _need(boxes, "boxes")
    img_of_box = _need(img_of_box, "img_of_box", torch.int32)
    rotation_deg = _need(rotation_deg, "rotation_deg")
    out = torch.empty_like(boxes)
    flipped = torch.empty_like(boxes) if flip_all else None
    with torch.cuda.device(boxes.device):
        st = lib.eqa_boxes_action(boxes.data_ptr(), img_of_box.data_ptr(), rotation_deg.data_ptr(),
                                  flipped.data_ptr() if flip_all else None, out.data_ptr(), boxes.shape[0], float(width),
                                  int(flip_all), _stream())
    _lib.check(st, "eqa_boxes_action")
    return out, flipped


def conv_s2_supported(cin: int, cout: int, k: int, pad: int, planar: bool) -> bool:
    return bool(_lib.load().eqa_conv_s2_supported(cin, cout, k, pad, int(planar)))


def pack_conv_s2_weights(w: torch.Tensor, planar: bool) -> torch.Tensor:
    """(Cout, Cin, K, K) filters -> the operand order of eqa_conv_s2 (layouts: include/eqa_hip.h)."""
    Cout, Cin, K, _ = w.shape
    if planar:
        wp = torch.zeros(Cout, 4, K, K, dtype=w.dtype, device=w.device)
        wp[:, :Cin] = w
        return wp.permute(2, 3, 1, 0).reshape(K * K, 4, Cout // 16, 16).permute(0, 2, 1, 3).contiguous()
    t = w.permute(2, 3, 1, 0).reshape(K * K, Cin // 16, 4, 4, Cout // 16, 16)      # tap, chunk, kq, s, n, j
    return t.permute(0, 1, 4, 2, 5, 3).contiguous()                                  # tap, chunk, n, kq, j, s


def conv_s2(x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], gelu: bool, cout: int, k: int, pad: int,
            planar: bool) -> torch.Tensor:
    """Stride-2 k x k convolution (+ bias, + exact GELU) on the fp32 MFMA (eqa_conv_s2).  planar: x (B,Cin<=4,H,W) contiguous;
    else x is a (B,H,W,Cin) contiguous tensor.  Returns (B,OH,OW,Cout) contiguous (channels-last data)."""
    lib = _lib.load()
    x, wp = _need(x, "x"), _need(wp, "wp")
    bias, p_bias = _opt(bias, "bias", torch.float32)
    if planar:
        B, cin, H, W = x.shape
    else:
        B, H, W, cin = x.shape
    OH, OW = (H + 2 * pad - k) // 2 + 1, (W + 2 * pad - k) // 2 + 1
    y = torch.empty((B, OH, OW, cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed("conv_s2"):
        st = lib.eqa_conv_s2(x.data_ptr(), wp.data_ptr(), p_bias, int(gelu), y.data_ptr(), B, cin, H, W, cout, k, pad, int(planar), _stream())
    _lib.check(st, "eqa_conv_s2")
    return y


def pack_conv_s2_dgrad_weights(w: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, K, K) filters -> the operand order of eqa_conv_s2_dgrad: (K*K, Cout/16, Cin/16, 4 [kq], 16 [j], 4 [s]) holding
    w[16 cc + 4 kq + s][16 n + j][u][v]."""
    Cout, Cin, K, _ = w.shape
    t = w.permute(2, 3, 0, 1).reshape(K * K, Cout // 16, 4, 4, Cin // 16, 16)      # tap, cc, kq, s, n, j
    return t.permute(0, 1, 4, 2, 5, 3).contiguous()                                  # tap, cc, n, kq, j, s


def conv_s2_train_supported(cin: int, cout: int, k: int, pad: int, planar: bool) -> bool:
    lib = _lib.load()
    return bool(lib.eqa_conv_s2_wgrad_supported(cin, cout, k, pad, int(planar))) and (planar or bool(lib.eqa_conv_s2_dgrad_supported(cin, cout, k, pad)))


def conv_s2_wgrad(x: torch.Tensor, dz: torch.Tensor, k: int, pad: int, planar: bool) -> torch.Tensor:
    """Filter gradient of eqa_conv_s2 (eqa_conv_s2_wgrad): x the layer's input (planar: (B,Cin,H,W); else (B,H,W,Cin)),
    dz (B,OH,OW,Cout) -> (Cout,Cin,K,K)."""
    lib = _lib.load()
    x, dz = _need(x, "x"), _need(dz, "dz")
    if planar:
        B, cin, H, W = x.shape
    else:
        B, H, W, cin = x.shape
    cout = dz.shape[-1]
    dw = torch.empty((cout, cin, k, k), dtype=torch.float32, device=x.device)
    ws = torch.empty((max(lib.eqa_conv_s2_wgrad_workspace_bytes(B, cin, H, W, cout, k, pad, int(planar)), 16) // 4,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed("conv_s2_wgrad"):
        st = lib.eqa_conv_s2_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, cin, H, W, cout, k, pad, int(planar), _stream())
    _lib.check(st, "eqa_conv_s2_wgrad")
    return dw


def conv_s2_dgrad(dz: torch.Tensor, wd: torch.Tensor, in_hw: Tuple[int, int], cin: int, k: int, pad: int) -> torch.Tensor:
    """Data gradient of eqa_conv_s2 for a channels-last layer (eqa_conv_s2_dgrad): dz (B,OH,OW,Cout), wd from
    ``pack_conv_s2_dgrad_weights`` -> (B,H,W,Cin)."""
    lib = _lib.load()
    dz, wd = _need(dz, "dz"), _need(wd, "wd")
    B, cout = dz.shape[0], dz.shape[-1]
    dx = torch.empty((B, in_hw[0], in_hw[1], cin), dtype=torch.float32, device=dz.device)
    with torch.cuda.device(dz.device), _timed("conv_s2_dgrad"):
        st = lib.eqa_conv_s2_dgrad(dz.data_ptr(), wd.data_ptr(), dx.data_ptr(), B, cin, in_hw[0], in_hw[1], cout, k, pad, _stream())
    _lib.check(st, "eqa_conv_s2_dgrad")
    return dx


def _bn_act_partials(z: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    npix, C = z.shape
    part = torch.empty((max(lib.eqa_bn_act_partial_blocks(npix), 1), C, 2), dtype=torch.float64, device=z.device)
    with torch.cuda.device(z.device), _timed("bn_act_stats"):
        st = lib.eqa_bn_act_stats(z.data_ptr(), part.data_ptr(), npix, C, _stream())
    _lib.check(st, "eqa_bn_act_stats")
    return part


def bn_batch_stats(z: torch.Tensor):
    """Per-channel batch mean and BIASED variance (fp64) of a channels-last (npix, C) view (eqa_bn_act_stats: fp32 sums inside
    blocks of up to 1024 pixels -- shifted by the block's first pixel, so that a large mean does not cancel the variance --, fp64 across them)."""
    z = _need(z, "z")
    sums = _bn_act_partials(z).sum(0)
    mean = sums[:, 0] / z.shape[0]
    var = (sums[:, 1] / z.shape[0] - mean * mean).clamp_min(0.0)
    return mean, var


def bn_fold_batch_stats(z: torch.Tensor, bn: torch.nn.modules.batchnorm._BatchNorm):
    """Batch statistics of a channels-last (npix, C) view folded with the module's affine parameters, the module's running
    statistics updated as a training-mode forward of nn.BatchNorm*d does -- eqa_bn_act_stats + ONE eqa_bn_act_finalize launch
    (the same through torch: ~25 element-wise launches on C-sized tensors) -> (scale, shift, mean, rstd) fp32."""
    from equiadapt_amd.common.utils import mark_written

    lib = _lib.load()
    z = _need(z, "z")
    npix, C = z.shape
    part = _bn_act_partials(z)
    out = torch.empty((4, C), dtype=torch.float32, device=z.device)
    track = bn.track_running_stats and bn.running_mean is not None
    momentum = 0.0
    if track:
        with torch.no_grad():
            bn.num_batches_tracked += 1
        momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
    with torch.cuda.device(z.device):
        st = lib.eqa_bn_act_finalize(part.data_ptr(), npix, C, bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps), float(momentum),
                                     bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
                                     out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), _stream())
    _lib.check(st, "eqa_bn_act_finalize")
    if track:
        mark_written(bn.running_mean, bn.running_var)
    return out[0], out[1], out[2], out[3]


def bn_act_fwd(z: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, rowscale: Optional[torch.Tensor], act: int) -> torch.Tensor:
    """y = rowscale[p] * act(scale[c] z + shift[c]) on a channels-last (npix, C) view (eqa_bn_act_fwd); act 0 = GELU (erf), 1 = ReLU."""
    lib = _lib.load()
    z, scale, shift = _need(z, "z"), _need(scale, "scale"), _need(shift, "shift")
    rowscale, p_rs = _opt(rowscale, "rowscale", torch.float32)
    npix, C = z.shape
    y = torch.empty_like(z)
    with torch.cuda.device(z.device), _timed("bn_act_fwd"):
        st = lib.eqa_bn_act_fwd(z.data_ptr(), scale.data_ptr(), shift.data_ptr(), p_rs, y.data_ptr(), npix, C, act, _stream())
    _lib.check(st, "eqa_bn_act_fwd")
    return y


def bn_act_bwd(gy: torch.Tensor, z: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
               gamma: torch.Tensor, rowscale: Optional[torch.Tensor], act: int):
    """Backward of batch-norm (batch statistics) + activation on (npix, C): -> (dz, dgamma, dbeta) (eqa_bn_act_bwd_reduce / _apply)."""
    lib = _lib.load()
    gy, z = _need(gy, "gy"), _need(z, "z")
    rowscale, p_rs = _opt(rowscale, "rowscale", torch.float32)
    npix, C = z.shape
    part = torch.empty((max(lib.eqa_bn_act_partial_blocks(npix), 1), C, 2), dtype=torch.float64, device=z.device)
    with torch.cuda.device(z.device), _timed("bn_act_bwd_reduce"):
        st = lib.eqa_bn_act_bwd_reduce(gy.data_ptr(), z.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), p_rs,
                                       part.data_ptr(), npix, C, act, _stream())
    _lib.check(st, "eqa_bn_act_bwd_reduce")
    coef = torch.empty((5, C), dtype=torch.float32, device=z.device)          # dgamma, dbeta, gscale, m1, m2
    with torch.cuda.device(z.device):
        st = lib.eqa_bn_act_bwd_finalize(part.data_ptr(), npix, C, gamma.data_ptr(), rstd.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                                         coef[2].data_ptr(), coef[3].data_ptr(), coef[4].data_ptr(), _stream())
    _lib.check(st, "eqa_bn_act_bwd_finalize")
    dgamma, dbeta, gscale, m1, m2 = coef[0], coef[1], coef[2], coef[3], coef[4]
    dz = torch.empty_like(z)
    with torch.cuda.device(z.device), _timed("bn_act_bwd_apply"):
        st = lib.eqa_bn_act_bwd_apply(gy.data_ptr(), z.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), p_rs,
                                      gscale.data_ptr(), m1.data_ptr(), m2.data_ptr(), dz.data_ptr(), npix, C, act, _stream())
    _lib.check(st, "eqa_bn_act_bwd_apply")
    return dz, dgamma, dbeta


def affine_relu_rows(h: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """relu(h * scale[d] + shift[d]) on (rows, D) (eqa_affine_relu_rows): eval-mode BatchNorm1d + ReLU in one pass."""
    lib = _lib.load()
    h, scale, shift = _need(h, "h"), _need(scale, "scale"), _need(shift, "shift")
    rows, D = h.shape
    z = torch.empty_like(h)
    with torch.cuda.device(h.device):
        st = lib.eqa_affine_relu_rows(h.data_ptr(), scale.data_ptr(), shift.data_ptr(), z.data_ptr(), rows, D, _stream())
    _lib.check(st, "eqa_affine_relu_rows")
    return z


def cosine_group_activations(v: torch.Tensor, ref: torch.Tensor, num_group: int, eps: float = 1e-8) -> torch.Tensor:
    """(G*B, V) element-major embeddings, (V) reference -> (B, G) cosine similarities (eqa_cosine_group_activations)."""
    lib = _lib.load()
    v, ref = _need(v, "vector_out"), _need(ref.reshape(-1), "reference_vector")
    R, V = v.shape
    if R % num_group or ref.numel() != V:
        raise ValueError("cosine_group_activations expects (G*B, V) embeddings and a (V) reference vector")
    B = R // num_group
    act = torch.empty((B, num_group), dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        st = lib.eqa_cosine_group_activations(v.data_ptr(), ref.data_ptr(), act.data_ptr(), B, num_group, V, float(eps), _stream())
    _lib.check(st, "eqa_cosine_group_activations")
    return act


def bias_relu_nhwc_(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """In place x = relu(x + bias[c]) on a channels-last (B,C,H,W) tensor (eqa_bias_relu_nhwc)."""
    lib = _lib.load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("bias_relu_nhwc_ expects a channels-last fp32 tensor on the device")
    bias = _need(bias, "bias")
    B, C, H, W = x.shape
    with torch.cuda.device(x.device), _timed("bias_relu"):
        st = lib.eqa_bias_relu_nhwc(x.data_ptr(), bias.data_ptr(), B * H * W, C, _stream())
    _lib.check(st, "eqa_bias_relu_nhwc")
    return x


def window_sums_gemv(S: torch.Tensor, weff: torch.Tensor, scale: float, shift) -> torch.Tensor:
    """(B, K) fp64 window sums x (E, K) fp64 mean-response weights -> (B, E) fp32 = scale * S @ weff^T + shift
    (eqa_window_sums_gemv).  `shift` may be a 0-dim device tensor (the bias mean): it is read without a host sync only
    if it already is a Python float."""
    lib = _lib.load()
    S = _need(S, "S", torch.float64)
    weff = _need(weff, "weff", torch.float64)
    B, K = S.shape
    E = weff.shape[0]
    act = torch.empty((B, E), dtype=torch.float32, device=S.device)
    sh = shift if isinstance(shift, float) else 0.0
    with torch.cuda.device(S.device), _timed("sums_gemv"):
        st = lib.eqa_window_sums_gemv(S.data_ptr(), weff.data_ptr(), act.data_ptr(), B, K, E, float(scale), sh, _stream())
    _lib.check(st, "eqa_window_sums_gemv")
    if not isinstance(shift, float):
        act += shift.to(torch.float32)
    return act


def window_sums_gemv_bwd(dact: torch.Tensor, weff: torch.Tensor, S: torch.Tensor, scale: float, need_dS: bool, need_dW: bool):
    """Backward of `window_sums_gemv`: (dS (B,K) fp64 or None, dweff (E,K) fp64 or None) (eqa_window_sums_gemv_bwd)."""
    lib = _lib.load()
    dact = _need(dact, "dact")
    weff, S = _need(weff, "weff", torch.float64), _need(S, "S", torch.float64)
    B, K = S.shape
    E = weff.shape[0]
    dS = torch.empty_like(S) if need_dS else None
    dW = torch.empty_like(weff) if need_dW else None
    ws = torch.empty((max(lib.eqa_window_sums_gemv_bwd_workspace_bytes(K, E), 8) // 8,), dtype=torch.float64, device=S.device) if need_dW else None
    with torch.cuda.device(S.device), _timed("sums_gemv_bwd"):
        st = lib.eqa_window_sums_gemv_bwd(dact.data_ptr(), weff.data_ptr(), S.data_ptr(), dS.data_ptr() if need_dS else None,
                                          dW.data_ptr() if need_dW else None, ws.data_ptr() if need_dW else None, B, K, E, float(scale), _stream())
    _lib.check(st, "eqa_window_sums_gemv_bwd")
    return dS, dW


def window_grad_table(dS: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """Backward of the window sums as a class table: dS (B,C,k,k) fp64 -> (B, 2k-1, 2k-1, C) fp32 (eqa_window_grad_table)."""
    lib = _lib.load()
    dS = _need(dS, "dS", torch.float64)
    B, C, k, _ = dS.shape
    T = 2 * (k - 1) + 1
    table = torch.empty((B, T, T, C), dtype=torch.float32, device=dS.device)
    with torch.cuda.device(dS.device), _timed("window_grad_table"):
        st = lib.eqa_window_grad_table(dS.data_ptr(), table.data_ptr(), B, C, H, W, k, _stream())
    _lib.check(st, "eqa_window_grad_table")
    return table


def plane_gemm_supported(cin: int, cout: int) -> bool:
    return bool(_lib.load().eqa_plane_gemm_supported(cin, cout))


def pack_plane_gemm_weights(U: torch.Tensor) -> torch.Tensor:
    """(P, Cin, Cout) Winograd-domain filters -> the MFMA operand-fragment order of eqa_plane_gemm:
    Upk[p][s][n][b][32 h + i][t] = U[p][16 s + 8 b + 4 h + t][32 n + i]."""
    P, Cin, Cout = U.shape
    t = U.float().reshape(P, Cin // 16, 2, 2, 4, Cout // 32, 32)        # p, s, b, h, t, n, i
    return t.permute(0, 1, 5, 2, 3, 6, 4).contiguous()                   # p, s, n, b, h, i, t


def plane_gemm(V: torch.Tensor, Upk: torch.Tensor, M: torch.Tensor, tiles: int) -> None:
    """M[:tiles, a] = V[:tiles, a] . U[a] for every plane a (eqa_plane_gemm); V:(T,P,Cin), M:(T,P,Cout) contiguous."""
    lib = _lib.load()
    V, Upk, M = _need(V, "V"), _need(Upk, "Upk"), _need(M, "M")
    P, Cin = V.shape[1], V.shape[2]
    Cout = M.shape[2]
    with torch.cuda.device(V.device):
        st = lib.eqa_plane_gemm(V.data_ptr(), Upk.data_ptr(), M.data_ptr(), tiles, P, Cin, Cout, _stream())
    _lib.check(st, "eqa_plane_gemm")


def lift_conv_supported(cin: int, kh: int, kw: int, cout: int) -> bool:
    """Shapes eqa_lift_conv_nhwc takes (others: the framework's convolution)."""
    return kh in (3, 5) and 9 <= kw * cin <= 15 and cout % 16 == 0


def pack_lift_weights(bank: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, KH, KW) filters -> (KH*8, 2, Cout) operand of eqa_lift_conv_nhwc (layout: include/eqa_hip.h): every
    filter row of R = KW*Cin floats is split into two 8-element halves starting at 0 and R-8; the elements the halves
    share get weight 0 in the second one."""
    Cout, Cin, KH, KW = bank.shape
    R = KW * Cin
    rows = bank.permute(2, 3, 1, 0).reshape(KH, R, Cout)                # [ky][j = kx*Cin + ci][co]
    q = torch.arange(8, device=bank.device)
    first = rows[:, q]                                                  # (KH, 8, Cout)
    second = rows[:, (R - 8) + q].clone()
    second[:, : 16 - R] = 0
    return torch.stack([first, second], dim=2).reshape(KH * 8, 2, Cout).float().contiguous()


def lift_conv_nhwc(x: torch.Tensor, wpk: torch.Tensor, bias: Optional[torch.Tensor], relu: bool, kh: int, kw: int) -> torch.Tensor:
    """Channels-last (B,Cin,H,W) -> channels-last (B,Cout,H-kh+1,W-kw+1) = [relu](conv2d(x, w) + bias) on the fp32 MFMA
    (eqa_lift_conv_nhwc); wpk = pack_lift_weights(w)."""
    lib = _lib.load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("lift_conv_nhwc expects a channels-last fp32 tensor on the device")
    wpk = _need(wpk, "wpk")
    bias, p_bias = _opt(bias, "bias", torch.float32)
    B, Cin, H, W = x.shape
    Cout = wpk.shape[2]
    y = torch.empty((B, Cout, H - kh + 1, W - kw + 1), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device), _timed("lift_conv"):
        st = lib.eqa_lift_conv_nhwc(x.data_ptr(), wpk.data_ptr(), p_bias, int(relu), y.data_ptr(), B, H, W, Cin, kh, kw, Cout, _stream())
    _lib.check(st, "eqa_lift_conv_nhwc")
    return y


def lift_conv_wide_supported(cin: int, kh: int, kw: int, cout: int) -> bool:
    """Shapes eqa_lift_conv_wide takes: the lifting filters eqa_lift_conv_nhwc does not (7 x 7 / 9 x 9 over RGB, 3 ... 9 over one channel)."""
    return bool(_lib.load().eqa_lift_conv_wide_supported(cin, kh, kw, cout))


def pack_lift_weights_wide(bank: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, KH, KW) filters -> (Cout/64, (R+1)/2, 2, 64) operand of eqa_lift_conv_wide (layout: include/eqa_hip.h): tap
    r = (ky*KW + kx)*Cin + ci, two taps per matrix instruction, a zero tap where R is odd."""
    Cout, Cin, KH, KW = bank.shape
    R = KH * KW * Cin
    steps = (R + 1) // 2
    taps = bank.permute(0, 2, 3, 1).reshape(Cout, R).float()            # [co][r]
    if 2 * steps > R:
        taps = torch.cat([taps, taps.new_zeros(Cout, 2 * steps - R)], dim=1)
    # [slice][nt][c32][step][khalf] -> [slice][step][nt][khalf][c32]
    t = taps.view(Cout // 64, 2, 32, steps, 2).permute(0, 3, 1, 4, 2)
    return t.reshape(Cout // 64, steps, 2, 64).contiguous()


def lift_conv_wide(x: torch.Tensor, wpk: torch.Tensor, bias: Optional[torch.Tensor], relu: bool, kh: int, kw: int) -> torch.Tensor:
    """`lift_conv_nhwc` for the wide / single-channel lifting filters (eqa_lift_conv_wide); wpk = pack_lift_weights_wide(w)."""
    lib = _lib.load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("lift_conv_wide expects a channels-last fp32 tensor on the device")
    wpk = _need(wpk, "wpk")
    bias, p_bias = _opt(bias, "bias", torch.float32)
    B, Cin, H, W = x.shape
    Cout = wpk.shape[0] * 64
    y = torch.empty((B, Cout, H - kh + 1, W - kw + 1), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device), _timed("lift_conv"):
        st = lib.eqa_lift_conv_wide(x.data_ptr(), wpk.data_ptr(), p_bias, int(relu), y.data_ptr(), B, H, W, Cin, kh, kw, Cout, _stream())
    _lib.check(st, "eqa_lift_conv_wide")
    return y


def lift_conv_wide_wgrad(x: torch.Tensor, dy: torch.Tensor, kh: int, kw: int) -> torch.Tensor:
    """d loss / d filters (Cout, Cin, kh, kw) of `lift_conv_wide` from channels-last x and dy, deterministic (eqa_lift_conv_wide_wgrad)."""
    lib = _lib.load()
    for t, name in ((x, "x"), (dy, "dy")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(memory_format=torch.channels_last)):
            raise RuntimeError(f"lift_conv_wide_wgrad expects a channels-last fp32 tensor on the device for {name}")
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    if tuple(dy.shape) != (B, Cout, H - kh + 1, W - kw + 1):
        raise RuntimeError("lift_conv_wide_wgrad: dy does not match x and the kernel size")
    ws = torch.empty(max(lib.eqa_lift_conv_wide_wgrad_workspace_bytes(Cin, kh, kw, Cout), 16) // 4, dtype=torch.float32, device=x.device)
    dbank = torch.empty((Cout, Cin, kh, kw), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = lib.eqa_lift_conv_wide_wgrad(x.data_ptr(), dy.data_ptr(), ws.data_ptr(), dbank.data_ptr(), B, H, W, Cin, kh, kw, Cout, _stream())
    _lib.check(st, "eqa_lift_conv_wide_wgrad")
    return dbank


def lift_conv_stats_supported(x_shape, kh: int, kw: int, cout: int) -> bool:
    """Shapes eqa_lift_conv_nhwc_stats takes (x_shape: (B, Cin, H, W))."""
    B, Cin, H, W = x_shape
    return _lib.load().eqa_lift_conv_stats_rows(B, H, W, Cin, kh, kw, cout) > 0


def lift_conv_nhwc_stats(x: torch.Tensor, wpk: torch.Tensor, kh: int, kw: int):
    """`lift_conv_nhwc` without bias / activation that also returns the fp64 partial sums (rows, Cout, 2) of the output's
    per-channel sum and sum of squares (eqa_lift_conv_nhwc_stats) -- what the batch-norm behind the lifting layer needs, taken
    from the values on their way out of the kernel.  None as the second result: this shape has no such form."""
    lib = _lib.load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("lift_conv_nhwc_stats expects a channels-last fp32 tensor on the device")
    wpk = _need(wpk, "wpk")
    B, Cin, H, W = x.shape
    Cout = wpk.shape[2]
    rows = lib.eqa_lift_conv_stats_rows(B, H, W, Cin, kh, kw, Cout)
    if rows <= 0:
        return lift_conv_nhwc(x, wpk, None, False, kh, kw), None
    y = torch.empty((B, Cout, H - kh + 1, W - kw + 1), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    part = torch.empty((rows, Cout, 2), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device), _timed("lift_conv"):
        st = lib.eqa_lift_conv_nhwc_stats(x.data_ptr(), wpk.data_ptr(), y.data_ptr(), part.data_ptr(), B, H, W, Cin, kh, kw, Cout, _stream())
    _lib.check(st, "eqa_lift_conv_nhwc_stats")
    return y, part


def lift_conv_grouped(x: torch.Tensor, wpk: torch.Tensor, bias: Optional[torch.Tensor], relu: bool, kh: int, kw: int) -> torch.Tensor:
    """`lift_conv_nhwc` with the output in the channel-group-major layout (B, Cout/16, H-kh+1, W-kw+1, 16) that the FFT
    convolution's input transform reads in whole cache lines (eqa_lift_conv_grouped)."""
    lib = _lib.load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("lift_conv_grouped expects a channels-last fp32 tensor on the device")
    wpk = _need(wpk, "wpk")
    bias, p_bias = _opt(bias, "bias", torch.float32)
    B, Cin, H, W = x.shape
    Cout = wpk.shape[2]
    y = torch.empty((B, Cout // 16, H - kh + 1, W - kw + 1, 16), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed("lift_conv"):
        st = lib.eqa_lift_conv_grouped(x.data_ptr(), wpk.data_ptr(), p_bias, int(relu), y.data_ptr(), B, H, W, Cin, kh, kw, Cout, _stream())
    _lib.check(st, "eqa_lift_conv_grouped")
    return y


def lift_conv_wgrad_supported(x: torch.Tensor, cout: int, kh: int, kw: int) -> bool:
    """Shapes eqa_lift_conv_wgrad_nhwc takes (others: the framework's convolution-weight-gradient)."""
    B, Cin, H, W = x.shape
    return bool(_lib.load().eqa_lift_conv_wgrad_supported(B, H, W, Cin, cout, kh, kw))


def lift_conv_wgrad_nhwc(x: torch.Tensor, dy: torch.Tensor, kh: int, kw: int) -> torch.Tensor:
    """d loss / d filters (Cout, Cin, kh, kw) of y = conv2d(x, w) from channels-last x (B,Cin,H,W) and dy (B,Cout,H-kh+1,W-kw+1)
    on the fp32 MFMA, deterministic (eqa_lift_conv_wgrad_nhwc)."""
    lib = _lib.load()
    for t, name in ((x, "x"), (dy, "dy")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(memory_format=torch.channels_last)):
            raise RuntimeError(f"lift_conv_wgrad_nhwc expects a channels-last fp32 tensor on the device for {name}")
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    if tuple(dy.shape) != (B, Cout, H - kh + 1, W - kw + 1):
        raise RuntimeError("lift_conv_wgrad_nhwc: dy does not match x and the kernel size")
    ws = torch.empty(max(lib.eqa_lift_conv_wgrad_workspace_bytes(B, H, W, Cin, Cout, kh, kw), 16) // 4, dtype=torch.float32, device=x.device)
    dbank = torch.empty((Cout, Cin, kh, kw), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = lib.eqa_lift_conv_wgrad_nhwc(x.data_ptr(), dy.data_ptr(), ws.data_ptr(), dbank.data_ptr(), B, H, W, Cin, Cout, kh, kw, _stream())
    _lib.check(st, "eqa_lift_conv_wgrad_nhwc")
    return dbank


def image_action_nearest(x: torch.Tensor, eidx: torch.Tensor, rtheta: torch.Tensor, flags: Optional[torch.Tensor],
                         pad: int, out_hw: Tuple[int, int], top_left: Tuple[int, int], n_planes: int, src_mod: int) -> torch.Tensor:
    """fp32 planes (P,H,W) -> (n_planes, OH, OW): nearest-neighbour action with edge pad + crop (eqa_image_action_nearest)."""
    lib = _lib.load()
    x = _need(x, "x")
    eidx = _need(eidx, "eidx", torch.int32)
    rtheta = _need(rtheta, "rtheta")
    flags, p_flags = _opt(flags, "flags", torch.int32)
    P, H, W = x.shape
    out = torch.empty((n_planes, out_hw[0], out_hw[1]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = lib.eqa_image_action_nearest(x.data_ptr(), out.data_ptr(), eidx.data_ptr(), rtheta.data_ptr(), p_flags,
                                          rtheta.shape[0], n_planes, src_mod, H, W, pad, out_hw[0], out_hw[1],
                                          top_left[0], top_left[1], _stream())
    _lib.check(st, "eqa_image_action_nearest")
    return out


def gram_schmidt_backward(v: torch.Tensor, grad_out: torch.Tensor) -> torch.Tensor:
    """dL/dv of `gram_schmidt` from dL/d(out), both (B,3,3) (eqa_gram_schmidt_bwd)."""
    lib = _lib.load()
    v = _need(v, "vectors")
    grad_out = _need(grad_out, "grad_out")
    if v.dim() != 3 or v.shape[1:] != (3, 3) or grad_out.shape != v.shape:
        raise ValueError("gram_schmidt_backward expects two (B,3,3) tensors")
    gv = torch.empty_like(v)
    with torch.cuda.device(v.device):
        st = lib.eqa_gram_schmidt_bwd(v.data_ptr(), grad_out.data_ptr(), gv.data_ptr(), v.shape[0], _stream())
    _lib.check(st, "eqa_gram_schmidt_bwd")
    return gv


def modified_gram_schmidt(v: torch.Tensor) -> torch.Tensor:
    """(B,3,3) -> orthonormal rows by MODIFIED Gram-Schmidt (n-body canonicalizer; eqa_modified_gram_schmidt)."""
    lib = _lib.load()
    v = _need(v, "vectors")
    out = torch.empty_like(v)
    with torch.cuda.device(v.device):
        st = lib.eqa_modified_gram_schmidt(v.data_ptr(), out.data_ptr(), v.shape[0], _stream())
    _lib.check(st, "eqa_modified_gram_schmidt")
    return out


def rigid_rows(x: torch.Tensor, R: torch.Tensor, t: Optional[torch.Tensor], inverse: bool) -> torch.Tensor:
    """Per-row rigid action on (M,3) row vectors: x R + t (inverse=False) or x R^T - t R^T (inverse=True)."""
    lib = _lib.load()
    x = _need(x, "x")
    R = _need(R, "R")
    t, p_t = _opt(t, "t", torch.float32)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        st = lib.eqa_rigid_rows(x.data_ptr(), R.data_ptr(), p_t, out.data_ptr(), x.shape[0], int(inverse), _stream())
    _lib.check(st, "eqa_rigid_rows")
    return out
