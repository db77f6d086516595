"""ctypes binding of libeqa_hip.so (C ABI: include/eqa_hip.h).

The product path has no CPU fallback: if the shared library is missing or does not export a symbol,
importing an op raises.  ``build()`` compiles it in-tree with hipcc for gfx950 (cross-compiles
without a GPU); ``__graft_entry__.build()`` calls it.
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("EQA_LIB") or os.path.join(CSRC, "libeqa_hip.so")  # EQA_LIB: A/B a variant build
SOURCES = [os.path.join(CSRC, f) for f in ("group_action.hip", "pooling.hip", "batchnorm.hip", "winograd.hip", "lift_conv.hip", "lift_conv_wide.hip", "lift_wgrad.hip", "pointcloud.hip", "vnsmall_train.hip", "vnsmall_tail.hip", "fftconv.hip", "lift_fft.hip", "cgemm3m.hip", "cgemm3m_bf16.hip", "smallconv.hip", "planegemm.hip", "convnet_train.hip")]
HEADERS = [os.path.join(CSRC, "eqa_common.hpp"), os.path.join(CSRC, "vn_common.hpp"), os.path.join(CSRC, "fft48.inc"), os.path.join(CSRC, "fft_common.inc")]
INCLUDE = os.path.join(ROOT, "include")

ABI_VERSION = 3   # == EQA_ABI_VERSION of include/eqa_hip.h (tests/test_abi_and_host.py compares the two)

_c_f = ctypes.POINTER(ctypes.c_float)
_c_i = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p
_int = ctypes.c_int

# name -> (restype, argtypes); must list every symbol include/eqa_hip.h declares (tests check this).
SIGNATURES = {
    "eqa_abi_version": (_int, []),
    "eqa_set_option": (_int, [_int, _int]),
    "eqa_get_option": (_int, [_int]),
    "eqa_fold_edge_pad_workspace_bytes": (ctypes.c_int64, [_int] * 4),
    "eqa_fold_edge_pad": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "eqa_canon_transform_fwd": (_int, [_vp, _vp, _vp, _vp, _vp] + [_int] * 6 + [_vp]),
    "eqa_invert_action_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [_int] * 6 + [_vp]),
    "eqa_group_action_pair": (_int, [_vp] * 4 + [_int, _int] + [_vp] * 5 + [_int, _int, _vp] + [_int] * 4 + [_vp]),
    "eqa_orbit_expand_fwd": (_int, [_vp, _vp, _vp, _vp] + [_int] * 5 + [_vp]),
    "eqa_group_action_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [_int] * 12 + [_vp]),
    "eqa_group_action_fwd_hint": (_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [_int] * 13 + [_vp]),
    "eqa_crop_resize_aa": (_int, [_vp] * 6 + [_int] * 9 + [_vp]),
    "eqa_mask_action_nearest": (_int, [_vp] * 5 + [_int] * 4 + [_vp]),
    "eqa_mask_action_nearest_planes": (_int, [_vp] * 5 + [_int] * 4 + [_vp]),
    "eqa_boxes_action": (_int, [_vp] * 5 + [_int, ctypes.c_float, _int, _vp]),
    "eqa_image_action_nearest": (_int, [_vp] * 5 + [_int] * 10 + [_vp]),
    "eqa_group_action_bwd_tiles": (_int, [_int, _int]),
    "eqa_group_action_bwd": (_int, [_vp] * 8 + [_int] * 12 + [_vp]),
    "eqa_group_action_bwd_theta": (_int, [_vp] * 8 + [_int] * 12 + [_vp]),
    "eqa_group_pool_workspace_bytes": (ctypes.c_int64, [_int] * 4),
    "eqa_group_pool_argmax": (_int, [_vp, _vp, _vp, _vp] + [_int] * 4 + [_vp]),
    "eqa_window_sums": (_int, [_vp, _vp, _vp, _int, _vp] + [_int] * 5 + [_vp]),
    "eqa_bn_partial_blocks": (ctypes.c_int64, [ctypes.c_int64]),
    "eqa_bn_stats_nhwc": (_int, [_vp, _vp, ctypes.c_int64, _int, _vp]),
    "eqa_bn_relu_dropout_nhwc": (_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, _int, ctypes.c_float, ctypes.c_uint32, _vp]),
    "eqa_bn_bwd_reduce_nhwc": (_int, [_vp] * 5 + [ctypes.c_float, _vp, ctypes.c_int64, _int, _vp, _vp, ctypes.c_uint32, _vp]),
    "eqa_bn_bwd_apply_nhwc": (_int, [_vp] * 8 + [ctypes.c_float, _vp, ctypes.c_int64, _int, _vp, _vp, ctypes.c_uint32, _vp]),
    "eqa_window_sums_nhwc_act": (_int, [_vp, _vp, _vp, _int, ctypes.c_float, ctypes.c_uint32, _vp, _vp] + [_int] * 5 + [_vp]),
    "eqa_bn_bwd_reduce_nhwc_wsgrad": (_int, [_vp] * 4 + [ctypes.c_float, _vp] + [_int] * 5 + [_vp, _vp, ctypes.c_uint32, _vp]),
    "eqa_bn_bwd_apply_nhwc_wsgrad": (_int, [_vp] * 7 + [ctypes.c_float, _vp] + [_int] * 5 + [_vp, _vp, ctypes.c_uint32, _vp]),
    "eqa_vn_blocks": (_int, [_int]),
    "eqa_vn_knn": (_int, [_vp, _vp, _int, _int, _int, _vp]),
    "eqa_vn_convpos_stats": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _vp]),
    "eqa_vn_convpos_fwd": (_int, [_vp] * 7 + [_int, _int, _int, _vp]),
    "eqa_vn_convpos_bwd_reduce": (_int, [_vp] * 10 + [_int, _int, _int, _vp]),
    "eqa_vn_convpos_bwd_apply": (_int, [_vp] * 12 + [_int, _int, _int, _vp]),
    "eqa_vn_tail_blocks": (_int, [_int]),
    "eqa_vn_tail_partial_floats": (_int, [_int]),
    "eqa_vn_tail_pass": (_int, [_int] + [_vp] * 8 + [_int, _int, _vp]),
    "eqa_vn_bn_finalize": (_int, [_vp, _int, _int, _int, ctypes.c_int64] + [_vp] * 5 + [ctypes.c_float, ctypes.c_float, _vp, _vp]),
    "eqa_vn_bn_bwd_finalize": (_int, [_vp, _int, _int, _int, ctypes.c_int64, _vp, _vp, _vp]),
    "eqa_window_grad_table": (_int, [_vp, _vp] + [_int] * 5 + [_vp]),
    "eqa_window_sums_gemv_bwd_workspace_bytes": (ctypes.c_int64, [_int, _int]),
    "eqa_window_sums_gemv_bwd": (_int, [_vp] * 6 + [_int, _int, _int, ctypes.c_double, _vp]),
    "eqa_window_sums_gemv": (_int, [_vp, _vp, _vp, _int, _int, _int, ctypes.c_double, ctypes.c_double, _vp]),
    "eqa_lift_conv_nhwc": (_int, [_vp, _vp, _vp, _int, _vp] + [_int] * 7 + [_vp]),
    "eqa_lift_conv_grouped": (_int, [_vp, _vp, _vp, _int, _vp] + [_int] * 7 + [_vp]),
    "eqa_lift_conv_stats_rows": (_int, [_int] * 7),
    "eqa_lift_conv_nhwc_stats": (_int, [_vp, _vp, _vp, _vp] + [_int] * 7 + [_vp]),
    "eqa_lift_conv_wgrad_supported": (_int, [_int] * 7),
    "eqa_lift_conv_wgrad_workspace_bytes": (ctypes.c_int64, [_int] * 7),
    "eqa_lift_conv_wgrad_nhwc": (_int, [_vp, _vp, _vp, _vp] + [_int] * 7 + [_vp]),
    "eqa_plane_gemm_supported": (_int, [_int, _int]),
    "eqa_plane_gemm": (_int, [_vp, _vp, _vp, ctypes.c_longlong, _int, _int, _int, _vp]),
    "eqa_winograd_f2k5_input": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    "eqa_winograd_f2k5_output": (_int, [_vp, _vp, _int, _vp, _int, _int, _int, _int, _vp]),
    "eqa_winograd_f4k5_input": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    "eqa_winograd_f4k5_output": (_int, [_vp, _vp, _int, _vp, _int, _int, _int, _int, _vp]),
    "eqa_winograd_f2k5_input_padded": (_int, [_vp, _vp, _int, _int, _int, _int, _int, _vp]),
    "eqa_winograd_f4k5_input_padded": (_int, [_vp, _vp, _int, _int, _int, _int, _int, _vp]),
    "eqa_winograd_f2k5_output_adjoint": (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    "eqa_winograd_f4k5_output_adjoint": (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    "eqa_winograd_f2k5_output_sums_workspace_bytes": (ctypes.c_int64, [_int] * 4),
    "eqa_winograd_f2k5_output_sums": (_int, [_vp, _vp, _int, _vp, _vp] + [_int] * 5 + [_vp]),
    "eqa_winograd_f4k5_output_sums": (_int, [_vp, _vp, _int, _vp, _vp] + [_int] * 5 + [_vp]),
    "eqa_fft48k5_tiles": (ctypes.c_int64, [_int]),
    "eqa_fft48k5_tile_pitch": (ctypes.c_int64, [ctypes.c_int64]),
    "eqa_fft48k5_frequencies": (_int, []),
    "eqa_fft48k5_filter_spectra": (_int, [_vp, _vp, _int, _int, _int, _vp]),
    "eqa_fft48k5_input_grad": (_int, [_vp, _vp, _vp] + [_int] * 4 + [_vp]),
    "eqa_fft48k5_grad_transform": (_int, [_vp, _vp, _vp] + [_int] * 4 + [_vp]),
    "eqa_fft48k5_filter_grad": (_int, [_vp, _vp, _int, _int, _vp]),
    "eqa_fft48k5_filter_grad3m": (_int, [_vp, _vp, _int, _int, _vp]),
    "eqa_fft48k5_cgemm3m_supported": (_int, [_int, _int]),
    "eqa_fft48k5_spectra3m_floats": (ctypes.c_int64, [_int, _int]),
    "eqa_fft48k5_filter_spectra3m": (_int, [_vp, _vp, _int, _int, _int, _vp]),
    "eqa_fft48k5_cgemm3m": (_int, [_vp, _vp, _vp, ctypes.c_int64, _int, _int, _vp]),
    "eqa_lift_conv_wide_supported": (_int, [_int, _int, _int, _int]),
    "eqa_lift_conv_wide_weight_floats": (ctypes.c_int64, [_int, _int, _int, _int]),
    "eqa_lift_conv_wide": (_int, [_vp, _vp, _vp, _int, _vp, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "eqa_lift_conv_wide_wgrad_workspace_bytes": (ctypes.c_int64, [_int, _int, _int, _int]),
    "eqa_lift_conv_wide_wgrad": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "eqa_fft48k5_spectra3m_bf16_bytes": (ctypes.c_int64, [_int, _int]),
    "eqa_fft48k5_spectra3m_split": (_int, [_vp, _vp, _int, _int, _vp]),
    "eqa_fft48k5_cgemm3m_bf16x3": (_int, [_vp, _vp, _vp, ctypes.c_int64, _int, _int, _int, _vp]),
    "eqa_fft48k5_spectra3m_f16_bytes": (ctypes.c_int64, [_int, _int]),
    "eqa_fft48k5_spectra3m_split_f16": (_int, [_vp, _vp, _int, _int, ctypes.c_float, _vp]),
    "eqa_fft48k5_cgemm3m_f16x2": (_int, [_vp, _vp, _vp, ctypes.c_int64, _int, _int, _vp, _int, ctypes.c_float, _vp]),
    "eqa_fft48k5_wgrad3m_supported": (_int, [_int, _int]),
    "eqa_fft48k5_wgrad3m": (_int, [_vp, _vp, _vp, ctypes.c_int64, _int, _int, _vp]),
    "eqa_fft48k5_group": (_int, [_int, _int]),
    "eqa_fft48k5_workspace_bytes": (ctypes.c_int64, [_int] * 4),
    "eqa_fft48k5_input": (_int, [_vp, _vp, _vp, _vp] + [_int] * 5 + [_vp]),
    "eqa_fft48k5_input_grouped": (_int, [_vp, _vp, _vp, _vp] + [_int] * 5 + [_vp]),
    "eqa_fft48k5_input_grouped_supported": (_int, [_int]),
    "eqa_lift5_fft48k5_input_supported": (_int, [_int] * 4),
    "eqa_lift5_fft48k5_input": (_int, [_vp, _vp, _vp, _int, _vp] + [_int] * 4 + [_vp]),
    "eqa_lift5_fft48k5_input_dcmax": (_int, [_vp, _vp, _vp, _int, _vp, _vp] + [_int] * 4 + [_vp]),
    "eqa_lift5_pieces_f16_bytes": (ctypes.c_int64, [_int]),
    "eqa_absmax_slots": (_int, [_vp, ctypes.c_int64, _vp, _vp]),
    "eqa_lift5_fft48k5_input_f16x2": (_int, [_vp, _vp, ctypes.c_float, _vp, _int, _vp, _int, _vp, _vp] + [_int] * 4 + [_vp]),
    "eqa_lift5_pieces_bytes": (ctypes.c_int64, [_int]),
    "eqa_lift5_fft48k5_input_bf16x3": (_int, [_vp, _vp, _vp, _int, _vp] + [_int] * 4 + [_vp]),
    "eqa_fft48k5_output": (_int, [_vp, _vp, _vp, _int, _vp] + [_int] * 4 + [_vp]),
    "eqa_fft48k5_output_stats_rows": (ctypes.c_int64, [_int] * 4),
    "eqa_fft48k5_output_stats": (_int, [_vp, _vp, _vp, _vp] + [_int] * 4 + [_vp]),
    "eqa_fft48k5_output_sums": (_int, [_vp, _vp, _vp, _int, _vp, _vp] + [_int] * 5 + [_vp]),
    "eqa_fft48_supported": (_int, [_int]),
    "eqa_fft48_tiles": (ctypes.c_int64, [_int, _int]),
    "eqa_fft48_workspace_bytes": (ctypes.c_int64, [_int] * 5),
    "eqa_fft48_filter_spectra": (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    "eqa_fft48_filter_spectra3m": (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    "eqa_fft48_input": (_int, [_vp, _vp, _vp, _vp] + [_int] * 6 + [_vp]),
    "eqa_fft48_grad_transform": (_int, [_vp, _vp, _vp] + [_int] * 5 + [_vp]),
    "eqa_fft48_output": (_int, [_vp, _vp, _vp, _int, _vp] + [_int] * 5 + [_vp]),
    "eqa_fft48_input_grad": (_int, [_vp, _vp, _vp] + [_int] * 5 + [_vp]),
    "eqa_fft48_filter_grad": (_int, [_vp, _vp, _int, _int, _int, _int, _vp]),
    "eqa_conv_s2_supported": (_int, [_int] * 5),
    "eqa_conv_s2": (_int, [_vp, _vp, _vp, _int, _vp] + [_int] * 8 + [_vp]),
    "eqa_conv_s2_wgrad_supported": (_int, [_int] * 5),
    "eqa_conv_s2_wgrad_workspace_bytes": (ctypes.c_int64, [_int] * 8),
    "eqa_conv_s2_wgrad": (_int, [_vp] * 4 + [_int] * 8 + [_vp]),
    "eqa_conv_s2_dgrad_supported": (_int, [_int] * 4),
    "eqa_conv_s2_dgrad": (_int, [_vp] * 3 + [_int] * 7 + [_vp]),
    "eqa_bn_act_fwd": (_int, [_vp] * 5 + [ctypes.c_int64, _int, _int, _vp]),
    "eqa_bn_act_partial_blocks": (ctypes.c_int64, [ctypes.c_int64]),
    "eqa_bn_act_stats": (_int, [_vp, _vp, ctypes.c_int64, _int, _vp]),
    "eqa_bn_act_finalize": (_int, [_vp, ctypes.c_int64, _int, _vp, _vp, ctypes.c_double, ctypes.c_double] + [_vp] * 6 + [_vp]),
    "eqa_bn_act_bwd_finalize": (_int, [_vp, ctypes.c_int64, _int] + [_vp] * 7 + [_vp]),
    "eqa_bn_act_bwd_reduce": (_int, [_vp] * 8 + [ctypes.c_int64, _int, _int, _vp]),
    "eqa_bn_act_bwd_apply": (_int, [_vp] * 11 + [ctypes.c_int64, _int, _int, _vp]),
    "eqa_affine_relu_rows": (_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, _int, _vp]),
    "eqa_cosine_group_activations": (_int, [_vp, _vp, _vp, _int, _int, _int, ctypes.c_float, _vp]),
    "eqa_bias_relu_nhwc": (_int, [_vp, _vp, ctypes.c_int64, _int, _vp]),
    "eqa_window_sums_nhwc_workspace_bytes": (ctypes.c_int64, [_int] * 4),
    "eqa_window_sums_nhwc": (_int, [_vp, _vp, _vp, _int, _vp, _vp] + [_int] * 5 + [_vp]),
    "eqa_window_sums_bwd_expand_nhwc": (_int, [_vp, _vp] + [_int] * 5 + [_vp]),
    "eqa_group_argmax": (_int, [_vp, _vp, _int, _int, _vp]),
    "eqa_vnsmall_workspace_bytes": (ctypes.c_int64, [_int, _int]),
    "eqa_vnsmall_fwd": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "eqa_vnsmall_canonicalize": (_int, [_vp] * 6 + [_int] * 4 + [_vp]),
    "eqa_so3_rotate": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp]),
    "eqa_gram_schmidt": (_int, [_vp, _vp, _int, _vp]),
    "eqa_gram_schmidt_bwd": (_int, [_vp, _vp, _vp, _int, _vp]),
    "eqa_modified_gram_schmidt": (_int, [_vp, _vp, _int, _vp]),
    "eqa_rigid_rows": (_int, [_vp, _vp, _vp, _vp, _int, _int, _vp]),
}

_lock = threading.Lock()
_lib = None


def source_hash(*names: str) -> str:
    """sha1 over the named csrc files (+ the shared headers): what a committed counter measurement of a kernel is stamped with
    (tools/collect_traffic*.sh) and what bench.py compares against before it reports that measurement beside a live timing."""
    import hashlib

    h = hashlib.sha1()
    for path in [os.path.join(CSRC, n) for n in names] + HEADERS:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


class EqaLibraryError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = None) -> str:
    """Compile csrc/*.hip -> csrc/libeqa_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    Each translation unit is compiled to its own object (in parallel, re-compiled only when it or a header changed) and the
    objects are linked; ``extra_flags`` / ``out`` build a variant (tools/ablate.sh, A/B runs through ``EQA_LIB``)."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor

    if extra_flags and not out:
        raise ValueError("a variant build (extra_flags) needs its own `out`: it must not replace the product library")
    out = out or SO_PATH
    hdr_time = max(os.path.getmtime(p) for p in HEADERS + [os.path.join(INCLUDE, "eqa_hip.h")])
    newest_src = max(hdr_time, max(os.path.getmtime(p) for p in SOURCES))
    if not force and not extra_flags and os.path.exists(out) and os.path.getmtime(out) >= newest_src:
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fno-slp-vectorize: left on, the SLP vectoriser packs adjacent fp32 adds / multiplies of the straight-line FFT and resampling
    # code into v_pk_*_f32 pairs and pays for them with register moves (956 v_mov in the fused inverse transform); measured
    # without it: inverse transform + window sums 0.90 -> 0.80 ms, fused VNSmall +5 %, group action 0.649 -> 0.671 of HBM peak
    flags = ["--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-I", INCLUDE, *extra_flags]
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    # One builder at a time per tree (round 6): torchrun ranks and pytest-xdist workers that meet a stale tree queue on this
    # lock; the first compiles and links, the others find the library fresh when their turn comes and return.  The post-link cleanup
    # below therefore never meets another process's in-flight temporaries (it still only removes what it can prove dead).
    import fcntl

    lock_file = open(os.path.join(objdir, ".lock"), "w")
    fcntl.flock(lock_file, fcntl.LOCK_EX)
    try:
        if not force and not extra_flags and os.path.exists(out) and os.path.getmtime(out) >= newest_src:
            return out
        return _build_locked(hipcc, flags, tag, objdir, out, hdr_time, force, verbose, bool(extra_flags))
    finally:
        fcntl.flock(lock_file, fcntl.LOCK_UN)
        lock_file.close()


def _pid_alive(pid: int) -> bool:
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    return True


def _build_locked(hipcc, flags, tag, objdir, out, hdr_time, force, verbose, variant):
    from concurrent.futures import ThreadPoolExecutor
    import re
    import time

    started = time.time()

    def compile_one(src):
        obj = os.path.join(objdir, f"{os.path.splitext(os.path.basename(src))[0]}.{tag}.o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(hdr_time, os.path.getmtime(src)):
            return obj, None
        # compile to a private name and rename: several processes building the same stale tree (torchrun ranks, pytest-xdist) never
        # see a half-written object, and an interrupted compile leaves no truncated file that passes the freshness check
        tmp_obj = f"{obj}.tmp{os.getpid()}"
        cmd = [hipcc, *flags, "-c", src, "-o", tmp_obj]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            if os.path.exists(tmp_obj):
                os.remove(tmp_obj)
            return obj, f"{' '.join(cmd)}\n{res.stdout}\n{res.stderr}"
        os.replace(tmp_obj, obj)
        return obj, None

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
        results = list(pool.map(compile_one, SOURCES))
    errors = [e for _, e in results if e]
    if errors:
        raise EqaLibraryError("hipcc failed:\n" + "\n".join(errors))
    tmp = out + f".tmp{os.getpid()}"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[o for o, _ in results], "-o", tmp]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise EqaLibraryError(f"link failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    os.replace(tmp, out)
    if not variant:
        # a product build leaves only the product's objects behind: the flag-hash variants of A/B experiments (tools/ablate.sh)
        # accumulate otherwise (round 4 ended with 23 stale sets, 104 MB).  Removed: finished objects of OTHER flag sets that are
        # older than this build's start, and compile temporaries whose owning process is gone -- never a live process's files.
        keep = {os.path.basename(o) for o, _ in results}
        for name in os.listdir(objdir):
            path = os.path.join(objdir, name)
            try:
                m = re.search(r"\.o\.tmp(\d+)$", name)
                if m:
                    if not _pid_alive(int(m.group(1))):
                        os.remove(path)
                elif name.endswith(".o") and name not in keep and os.path.getmtime(path) < started:
                    os.remove(path)
            except OSError:
                pass
    return out


def load() -> ctypes.CDLL:
    """dlopen the library (once) and type every entry point.  Raises if it is absent: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(SO_PATH):
            raise EqaLibraryError(
                f"{SO_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). equiadapt_amd has no CPU fallback."
            )
        lib = ctypes.CDLL(SO_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as exc:
                raise EqaLibraryError(f"{SO_PATH} does not export {name}; rebuild it") from exc
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.eqa_abi_version() != ABI_VERSION:
            raise EqaLibraryError("libeqa_hip.so ABI version mismatch; rebuild it")
        _lib = lib
    return _lib


_ERRORS = {-1: "invalid argument", -2: "kernel launch failed", -3: "unsupported size/shape"}


def check(status: int, what: str) -> None:
    if status != 0:
        raise EqaLibraryError(f"{what} failed: {_ERRORS.get(status, status)}")
