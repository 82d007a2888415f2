"""ctypes binding of libmbx.so (the C-ABI in include/mbx.h).

There is NO fallback: if the library is missing or a call fails, this raises.  Device
pointers are passed as integers (``tensor.data_ptr()``); the stream is
``torch.cuda.current_stream().cuda_stream``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmbx.so")

P = C.c_void_p
I = C.c_int
F = C.c_float
D = C.c_double
SZ = C.c_size_t


class PatchMeta(C.Structure):
    """mbx_patch_meta (include/mbx.h); the per-patch columns of detect.py:190-281."""
    _fields_ = [("offset_y", C.c_int32), ("offset_x", C.c_int32), ("patch_h", C.c_int32), ("patch_w", C.c_int32),
                ("image_h", C.c_int32), ("image_w", C.c_int32), ("is_flipped", C.c_int32),
                ("max_to_keep", C.c_int32), ("restrictions", C.c_float * 4)]


_SIGS = {
    "mbx_version": (I, []),
    "mbx_status_string": (C.c_char_p, [I]),
    "mbx_priors_count": (I, [I, P, I]),
    "mbx_crc32c": (C.c_uint32, [P, C.c_uint64, C.c_uint32]),
    "mbx_generate_priors": (I, [P, I, D, D, I, P, I, P]),
    "mbx_decode_conf": (I, [P, P, P, I, I, F, P, P, P]),
    "mbx_match_workspace_bytes": (SZ, [I, I, I]),
    "mbx_match": (I, [P, P, P, P, F, I, I, I, P, P, P, SZ, P]),
    "mbx_loss_workspace_bytes": (SZ, [I]),
    "mbx_loss_fwd_bwd": (I, [P, P, I, P, P, F, F, I, I, I, P, P, P, P, SZ, P]),
    "mbx_decode_filter_topk": (I, [P, P, P, P, I, I, I, P, P, P, P, P]),
    "mbx_nms": (I, [P, P, P, P, I, I, C.c_double, P]),
    "mbx_augment_workspace_bytes": (SZ, [I, I]),
    "mbx_augment_batch": (I, [P, P, I, I, I, P, P, P]),
    "mbx_extract_patches": (I, [P, P, I, I, P, P]),
    "mbx_conv_stats_rows": (I, [P]),
    "mbx_conv": (I, [P, P]),
    "mbx_conv_supported": (I, [P]),
    "mbx_conv_pair": (I, [P, P, P]),
    "mbx_conv_splitk_workspace_bytes": (SZ, [P]),
    "mbx_conv_wgrad": (I, [P, P, C.c_int64, I, P, P, P]),
    "mbx_conv_wgrad_scaled": (I, [P, P, C.c_int64, I, F, P, P, P]),
    "mbx_wgrad_plan_bytes": (SZ, [P, I, I]),
    "mbx_wgrad_plan": (I, [P, I, I, P, SZ, P]),
    "mbx_conv_wgrad_grouped": (I, [P, P, P]),
    "mbx_conv_wgrad_grouped_capped": (I, [P, P, I, P]),
    "mbx_bn_finalize": (I, [P, I, I, C.c_int64, F, F, P, P, P, P, P]),
    "mbx_bn_apply": (I, [P, C.c_int64, I, P, P, P, I, P, I, P]),
    "mbx_bn_moving_update": (I, [P, P, P, P, C.c_int64, F, P, P, P, P, F, P]),
    "mbx_bn_finalize_parts": (I, [P, P, P, I, C.c_int64, F, F, P, P, P, P, P]),
    "mbx_bn_apply_mapped": (I, [P, C.c_int64, I, P, P, P, I, P, I, P, P]),
    "mbx_bn_bwd_reduce_mapped": (I, [P, I, P, I, I, P, C.c_int64, I, P, P, P, P, P, P]),
    "mbx_bn_bwd_apply_mapped": (I, [P, I, P, I, I, P, C.c_int64, I, P, P, P, P, P, P, P]),
    "mbx_bn_apply_maxpool": (I, [P, I, I, I, I, P, P, P, I, P, C.c_int64, I, I, I, P, P]),
    "mbx_bn_bwd_rows_pooled": (I, [I, I, I, I]),
    "mbx_bn_bwd_reduce_pooled": (I, [P, C.c_int64, I, P, I, I, I, I, I, I, P, I, P, P, P, P, P]),
    "mbx_bn_bwd_apply_pooled": (I, [P, C.c_int64, I, P, I, I, I, I, I, I, P, I, P, P, P, P, P, P]),
    "mbx_bn_bwd_onepass_mapped": (I, [P, I, I, P, C.c_int64, I, P, P, P, P, P, P, I, P, P, P]),
    "mbx_bn_fold": (I, [P, P, P, F, I, P, P, P]),
    "mbx_bn_apply_fused": (I, [P, I, C.c_int64, F, F, P, C.c_int64, I, P, I, P, I, P, P, P, P, P]),
    "mbx_bn_apply_fused_mapped": (I, [P, I, C.c_int64, F, F, P, C.c_int64, I, P, I, P, I, P, P, P, P, P, P, P]),
    "mbx_bn_bwd_apply_rows": (I, [P, I, P, I, P, C.c_int64, I, P, P, P, P, P, P, P]),
    "mbx_bn_bwd_rows": (I, [C.c_int64, I]),
    "mbx_bn_bwd_onepass_workspace_bytes": (C.c_size_t, [I]),
    "mbx_bn_bwd_onepass_supported": (I, [C.c_int64, I, I]),
    "mbx_bn_bwd_onepass": (I, [P, I, I, P, C.c_int64, I, P, P, P, P, P, P, I, P, P]),
    "mbx_bn_bwd_reduce": (I, [P, I, P, I, I, P, C.c_int64, I, P, P, P, P, P]),
    "mbx_bn_bwd_finalize": (I, [P, I, I, C.c_int64, P, P, P]),
    "mbx_bn_bwd_apply": (I, [P, I, P, I, I, P, C.c_int64, I, P, P, P, P, P, P]),
    "mbx_maxpool_fwd": (I, [P, C.c_int64, I, I, I, I, I, I, I, P, C.c_int64, I, I, I, P, P]),
    "mbx_maxpool_bwd": (I, [P, C.c_int64, I, P, I, I, I, I, I, I, I, I, P, C.c_int64, I, I, P]),
    "mbx_avgpool_fwd": (I, [P, C.c_int64, I, I, I, I, I, I, I, P, C.c_int64, I, I, I, P]),
    "mbx_avgpool_bwd": (I, [P, C.c_int64, I, I, I, I, I, I, I, I, I, P, C.c_int64, I, I, P]),
    "mbx_relu_mask": (I, [P, I, P, I, C.c_int64, I, P]),
    "mbx_pack_input": (I, [P, C.c_int64, P, P]),
    "mbx_head_gather": (I, [P, I, I, I, I, I, I, P, P, P]),
    "mbx_head_scatter": (I, [P, P, I, I, I, I, I, P, I, P]),
    "mbx_head_gather_all": (I, [P, I, I, I, P, P, P]),
    "mbx_head_scatter_all": (I, [P, P, P, I, I, I, P]),
    "mbx_step_begin": (I, [P, C.c_int64, P, C.c_int64, C.c_int64, P, P, P]),
    "mbx_filter_prepare": (I, [P, P, P, I, I, P]),
    "mbx_rmsprop_ema_step": (I, [P, P, P, P, P, P, C.c_int64, F, F, F, F, F, F, I, P, P, P]),
    "mbx_ema_update": (I, [P, P, C.c_int64, F, P, P]),
}

_lib = None


class MbxError(RuntimeError):
    pass


class ChanMap(C.Structure):
    """mbx_chan_map (include/mbx.h)."""
    _fields_ = [("n", C.c_int32), ("c_begin", C.c_int32 * 4), ("offset", C.c_int32 * 4)]


class BnBwdStats(C.Structure):
    """mbx_bn_bwd_stats (include/mbx.h)."""
    _fields_ = [("n", C.c_int32), ("rows_mod", C.c_int32), ("c_begin", C.c_int32 * 4), ("y", C.c_void_p * 4), ("ld_y", C.c_int32 * 4),
                ("relu_thr", C.c_void_p * 4), ("stats", C.c_void_p * 4), ("stats_ld", C.c_int32 * 4)]


class Head(C.Structure):
    """mbx_head (include/mbx.h)."""
    _fields_ = [("h", C.c_void_p), ("ld_h", C.c_int32), ("g", C.c_void_p), ("ld_g", C.c_int32), ("cells", C.c_int32),
                ("k", C.c_int32), ("off", C.c_int32)]


class FilterEntry(C.Structure):
    """mbx_filter_entry (include/mbx.h)."""
    _fields_ = [("src_off", C.c_int64), ("dst_off", C.c_int64), ("K", C.c_int32), ("R", C.c_int32), ("S", C.c_int32),
                ("C", C.c_int32), ("Kpad", C.c_int32), ("first_block", C.c_int32)]


def lib():
    """Load libmbx.so once.  Raises (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MbxError("libmbx.so not built at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(needs hipcc); there is no CPU fallback" % LIB_PATH)
        # torch bundles its own libamdhip64.so.7; load it FIRST so libmbx binds to the same HIP
        # runtime instance (two runtimes in one process: "no ROCm-capable device is detected").
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)      # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().mbx_status_string(int(status))
        raise MbxError("%s failed: %s (%d)" % (what or "libmbx call", msg.decode() if msg else "?", status))


def declared_symbols():
    return sorted(_SIGS)
