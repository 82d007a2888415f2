"""Thin Python launchers over the libmbx convolution-stack entry points (include/mbx.h).

``View`` is an NHWC bf16 tensor *view*: a channel slice [ch_off, ch_off+C) of a buffer whose
pixel stride is ``ld`` -- how tf.concat(3, ...) of model.py is made free.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib

EPI_STORE, EPI_AFFINE, EPI_RESIDUAL, EPI_STORE_F32 = 0, 1, 2, 3


class ConvDesc(C.Structure):
    """mbx_conv_desc (include/mbx.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("x_img_stride", C.c_int64), ("ldx", C.c_int32),
        ("N", C.c_int32), ("H_in", C.c_int32), ("W_in", C.c_int32), ("C_in", C.c_int32),
        ("w", C.c_void_p), ("C_out", C.c_int32), ("R", C.c_int32), ("S", C.c_int32),
        ("stride", C.c_int32), ("transposed", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("H_out", C.c_int32), ("W_out", C.c_int32),
        ("y", C.c_void_p), ("y_img_stride", C.c_int64), ("ldy", C.c_int32),
        ("epilogue", C.c_int32), ("relu", C.c_int32), ("accumulate", C.c_int32),
        ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("skip", C.c_void_p), ("skip_img_stride", C.c_int64), ("ld_skip", C.c_int32), ("rscale", C.c_float),
        ("stats_partial", C.c_void_p),
        ("tile_config", C.c_int32),
        ("acc_src", C.c_void_p), ("acc_img_stride", C.c_int64), ("ld_acc", C.c_int32),
        ("work_counter", C.c_void_p),
        ("max_workgroups", C.c_int32),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64),
        ("stats_rows_mod", C.c_int32), ("stats_ld", C.c_int32),
        ("bn_bwd_stats", C.c_void_p),
        ("relu_bits", C.c_void_p), ("ld_bits", C.c_int32),
        ("bn_apply", C.c_void_p),
        ("bn_bwd", C.c_void_p),
    ]


class BnApplyDesc(C.Structure):
    """mbx_bn_apply_desc (include/mbx.h)."""
    _fields_ = [("barrier", C.c_void_p), ("a", C.c_void_p), ("ld_a", C.c_int32), ("beta", C.c_void_p), ("mean", C.c_void_p),
                ("rstd", C.c_void_p), ("moving_mean", C.c_void_p), ("moving_var", C.c_void_p), ("relu_thr", C.c_void_p),
                ("relu", C.c_int32), ("eps", C.c_float), ("decay", C.c_float), ("step_poison", C.c_void_p)]


class BnBwdFused(C.Structure):
    """mbx_bn_bwd_fused (include/mbx.h)."""
    _fields_ = [("barrier", C.c_void_p), ("n", C.c_int32), ("c_begin", C.c_int32 * 4), ("y", C.c_void_p * 4), ("ld_y", C.c_int32 * 4),
                ("dy", C.c_void_p * 4), ("ld_dy", C.c_int32 * 4), ("mean", C.c_void_p * 4), ("rstd", C.c_void_p * 4),
                ("beta", C.c_void_p * 4), ("dbeta", C.c_void_p * 4), ("acc", C.c_void_p * 4), ("acc_ld", C.c_int32 * 4),
                ("relu", C.c_int32 * 4), ("step_poison", C.c_void_p)]


GRID_BARRIER_BYTES = 6400                      # MBX_GRID_BARRIER_BYTES
BN_BWD_SLOTS = 8                               # MBX_BN_BWD_SLOTS


def i5_tile_for(M, C_out, n_cus=256):
    """An igemm5 tile (tile_config) for a launch that must be PERSISTENT (a grid-barrier tail) where the measured table chose a
    non-persistent tile: the tile with the most workgroups that still fit one round, else the fewest rounds (a tail walks a
    workgroup's tiles one after the other), then the least work per CU."""
    best, best_key = None, None
    for i, (bm, bn) in enumerate(I5_TILES):
        if (bm, bn) == (128, 192):
            continue                                # (no grid-barrier tail on that tile: csrc/conv5.hip)
        t = -(-M // bm) * -(-C_out // bn)
        key = (0, -t, 0) if t <= n_cus else (1, -(-t // n_cus), -(-t // n_cus) * bm * bn)
        if best_key is None or key < best_key:
            best, best_key = I5_FLAG + 1 + i, key
    return best


def fused_tail_rounds(desc, n_cus=256):
    """Tiles per workgroup of the persistent / one-tile-per-workgroup launch `desc` would be (resident image, whole-width direct,
    else the igemm5 tile with the fewest rounds): what a grid-barrier tail walks serially.  Measured (LAB_NOTES round 6): a tail
    on ONE tile costs about what the launch it replaces costs; on several tiles per workgroup it costs more."""
    M = desc.N * desc.H_out * desc.W_out
    if resident_applies(desc, min_images=1):
        return -(-desc.N * 4 // n_cus)
    if directw_applies(desc):
        th = max(1, 256 // desc.W_out)
        return -(-desc.N * -(-desc.H_out // th) // n_cus)
    return min(-(-(-(-M // bm) * -(-desc.C_out // bn)) // n_cus) for bm, bn in I5_TILES if (bm, bn) != (128, 192))


class WgradJob(C.Structure):
    """mbx_wgrad_job (include/mbx.h)."""
    _fields_ = [("desc", ConvDesc), ("dy", C.c_void_p), ("dy_img_stride", C.c_int64), ("ld_dy", C.c_int32),
                ("scale", C.c_float), ("dw", C.c_void_p), ("db", C.c_void_p)]


class WgradPlanInfo(C.Structure):
    """mbx_wgrad_plan_info (include/mbx.h)."""
    _fields_ = [("n_layers", C.c_int32), ("n_items", C.c_int32), ("layers_off", C.c_int64), ("items_off", C.c_int64),
                ("queues_off", C.c_int64), ("heads_off", C.c_int64), ("flops", C.c_double), ("tally_off", C.c_int64)]


class View:
    """Channel-slice view of an NHWC buffer [N, H, W, ld] (bf16 unless elem_size says otherwise)."""
    __slots__ = ("buf", "N", "H", "W", "C", "ld", "ch_off", "elem_size")

    def __init__(self, buf, N, H, W, C, ld=None, ch_off=0, elem_size=2):
        self.buf, self.N, self.H, self.W, self.C = buf, N, H, W, C
        self.ld = C if ld is None else ld
        self.ch_off, self.elem_size = ch_off, elem_size
        assert buf.numel() * buf.element_size() >= N * H * W * self.ld * elem_size, "view larger than its buffer"

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.ch_off * self.elem_size

    @property
    def img_stride(self):
        return self.H * self.W * self.ld

    @property
    def M(self):
        return self.N * self.H * self.W

    def slice(self, off, C):
        assert off + C <= self.C
        return View(self.buf, self.N, self.H, self.W, C, self.ld, self.ch_off + off, self.elem_size)

    def tensor(self):
        """Materialise as a [N,H,W,C] tensor (tests / debugging)."""
        dt = self.buf.dtype
        flat = self.buf.reshape(-1)[: self.N * self.H * self.W * self.ld].reshape(self.N, self.H, self.W, self.ld)
        return flat[..., self.ch_off:self.ch_off + self.C]

    @staticmethod
    def alloc(N, H, W, C, ld=None, dtype=torch.bfloat16, device="cuda", zero=False):
        ld = C if ld is None else ld
        f = torch.zeros if zero else torch.empty
        buf = f((N, H, W, ld), dtype=dtype, device=device)
        if not zero and os.environ.get("MBX_POISON"):      # debugging: make any read-before-write visible as NaN
            buf.fill_(float("nan"))
        return View(buf, N, H, W, C, ld, 0, torch.empty((), dtype=dtype).element_size())


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def make_desc(x: View, w, C_out, R, S, stride, pad_t, pad_l, y: View, transposed=0, epilogue=EPI_STORE, relu=0,
              accumulate=0, scale=None, shift=None, skip: View = None, rscale=0.0, stats=None, acc_src: View = None,
              stats_rows_mod=0, stats_ld=0, relu_bits=None, ld_bits=0):
    d = ConvDesc()
    d.x, d.x_img_stride, d.ldx = x.ptr, x.img_stride, x.ld
    d.N, d.H_in, d.W_in, d.C_in = x.N, x.H, x.W, x.C
    d.w, d.C_out, d.R, d.S = (w.data_ptr() if w is not None else None), C_out, R, S
    d.stride, d.transposed, d.pad_t, d.pad_l = stride, transposed, pad_t, pad_l
    d.H_out, d.W_out = y.H, y.W
    d.y, d.y_img_stride, d.ldy = y.ptr, y.img_stride, y.ld
    d.epilogue, d.relu, d.accumulate = epilogue, int(relu), int(accumulate)
    d.scale, d.shift = _p(scale), _p(shift)
    if skip is not None:
        d.skip, d.skip_img_stride, d.ld_skip = skip.ptr, skip.img_stride, skip.ld
    d.rscale = float(rscale)
    d.stats_partial = _p(stats)
    d.stats_rows_mod, d.stats_ld = int(stats_rows_mod), int(stats_ld)
    if acc_src is not None:
        d.acc_src, d.acc_img_stride, d.ld_acc = acc_src.ptr, acc_src.img_stride, acc_src.ld
    if relu_bits is not None:
        # sign bits of a residual output (uint8 tensor [M, ld_bits]): written by RESIDUAL + relu, read as the mask of a STORE
        d.relu_bits, d.ld_bits = relu_bits.data_ptr(), int(ld_bits if ld_bits else relu_bits.shape[-1])
    return d


def conv(desc: ConvDesc):
    _lib.check(_lib.lib().mbx_conv(C.byref(desc), _stream()), "mbx_conv")


def conv_stats_rows(desc: ConvDesc):
    r = _lib.lib().mbx_conv_stats_rows(C.byref(desc))
    if r < 0:
        _lib.check(r, "mbx_conv_stats_rows")
    return r


def conv_wgrad(desc: ConvDesc, dy: View, dw, db=None):
    _lib.check(_lib.lib().mbx_conv_wgrad(C.byref(desc), dy.ptr, dy.img_stride, dy.ld, dw.data_ptr(), _p(db), _stream()),
               "mbx_conv_wgrad")


# Overlapped weight gradients (Trainer): the grouped weight-gradient launch of backward segment i runs on a second stream
# with this many persistent workgroups (one CU each) BESIDE the data-gradient / batch-norm chain of segments i+1, ..., whose
# persistent launches (igemm5 / igemm7 / one-launch BN backward) are capped at the remaining CUs.  0 = off (the grouped
# launch runs at the end of its own segment, on every CU).  MBX_WG_OVERLAP overrides.
WG_OVERLAP_DEFAULT = 0


def wgrad_overlap_cus():
    return max(0, int(os.environ.get("MBX_WG_OVERLAP", WG_OVERLAP_DEFAULT)))


N_TILE_CONFIGS = 14
I5_FLAG = 32                                   # tile_config 32 + t: persistent igemm5 launch (csrc/conv5.hip), tile t
I5_TILE_CONFIGS = (33, 34, 35, 36, 37, 38, 39)     # 128x64, 128x128, 192x128, 256x128, 256x64, 128x192, 128x256
I5_TILES = ((128, 64), (128, 128), (192, 128), (256, 128), (256, 64), (128, 192), (128, 256))
I7_TILE_CONFIG = 65                            # persistent pointwise launch with the filter panel resident in LDS (csrc/conv7.hip)
I7_COUNTERS = 32                               # its work counters: one int per 128-channel column tile
SPLITK_FLAG = 128                              # tile_config 128 + S: split-K in S slices (float32 partials + reduce launch)
DIRECT3_TILE_CONFIG = 96                       # the direct 3x3 launch for few channels on large maps (csrc/convd.hip)


def direct3_applies(desc: ConvDesc, min_pixels=300000):
    """The direct 3x3 launch by rule: 3x3 / stride 1 (forward or data gradient), C_in 32 / 64, C_out <= 64, plain bf16 store
    (+ statistics) or the affine epilogue of a folded batch norm, and a map large enough that the nine-fold gather of the implicit GEMM is what the launch spends its
    time on (the stem's 147 x 147 layers at BATCH_SIZE 64: 1.4 M pixels).  MBX_DIRECT3=0 turns it off (A/B); MBX_DIRECT3_MIN_PIXELS."""
    if os.environ.get("MBX_DIRECT3", "1") == "0":
        return False
    if desc.R != 3 or desc.S != 3 or desc.epilogue not in (EPI_STORE, EPI_AFFINE) or desc.accumulate or desc.skip \
            or desc.rscale != 0.0 or (desc.epilogue == EPI_AFFINE and desc.stats_partial) or desc.bn_bwd_stats:
        return False
    if desc.stride == 2 and not desc.transposed and desc.C_in == 8 and desc.C_out <= 32 and desc.C_out % 8 == 0:
        pass                                       # the network's first layer (model.py:90): conv_stem_kernel, same tile_config
    elif desc.stride != 1 or desc.C_in not in (32, 64) or desc.C_out > 64 or desc.C_out % 8 or (desc.C_in == 64 and desc.C_out > 48):
        return False
    # its tiles are 8 rows x 32 columns: a map whose width fills the last column tile badly (35 -> 64: 1.8x the pixels) is
    # better off on the implicit GEMM (measured on block35's 3x3 layers, MBX_DIRECT3_MIN_PIXELS=70000: +0.06 ms per step)
    if -(-desc.W_out // 32) * 32 > 1.15 * desc.W_out:
        return False
    return desc.N * desc.H_out * desc.W_out >= int(os.environ.get("MBX_DIRECT3_MIN_PIXELS", min_pixels))


DIRECTW_TILE_CONFIG = 97                       # the whole-width direct 3x3 launch for narrow maps (csrc/convd.hip, conv_directw_kernel)


def directw_applies(desc: ConvDesc, min_pixels=60000):
    """The whole-width direct 3x3 launch by rule: 3x3 / stride 1 (forward or data gradient), C_in 32 / 48 / 64, C_out <= 64,
    plain bf16 store (+ statistics) or affine epilogue, a map 8..64 wide (block35's 35 x 35 layers: 78 400 pixels at
    BATCH_SIZE 64) and enough pixels to fill the chip.  MBX_DIRECTW=0 turns it off (A/B); MBX_DIRECTW_MIN_PIXELS."""
    if os.environ.get("MBX_DIRECTW", "1") == "0":
        return False
    if desc.R != 3 or desc.S != 3 or desc.stride != 1 or desc.epilogue not in (EPI_STORE, EPI_AFFINE) or desc.accumulate or desc.skip \
            or desc.rscale != 0.0 or (desc.epilogue == EPI_AFFINE and desc.stats_partial) or desc.bn_bwd_stats:
        return False
    if desc.C_in not in (32, 48, 64) or desc.C_out > 64 or desc.C_out % 8 or (desc.C_in == 64 and desc.C_out > 48):
        return False
    if not (8 <= desc.W_out <= 64) or desc.W_in > desc.W_out + 2:
        return False
    return desc.N * desc.H_out * desc.W_out >= int(os.environ.get("MBX_DIRECTW_MIN_PIXELS", min_pixels))


RESIDENT_TILE_CONFIG = 98                      # the resident-image launch for multi-tap convolutions on small maps (csrc/convr.hip)


def resident_applies(desc: ConvDesc, min_images=32):
    """The resident-image launch by rule: a stride-1, same-size convolution with a ONE-DIMENSIONAL multi-tap filter on a small
    map with many channels -- block17's 1x7 / 7x1 layers (17 x 17, C_in 128 / 160 / 192: model.py:33-37) and block8's 1x3 / 3x1
    (8 x 8, C_in 192 / 224 / 256: model.py:53-57), forward or data gradient: a tile = a whole image (staged in LDS once) x a
    quarter of the output channels, where the implicit GEMM gathers every pixel row once per tap.  Plain bf16 store
    (+ statistics) or the affine epilogue; enough images to fill the chip (4 tiles per image).  MBX_RESIDENT=0 turns it off
    (A/B).  The library has the last word: the rule asks mbx_conv_supported."""
    if os.environ.get("MBX_RESIDENT", "1") == "0":
        return False
    if desc.R * desc.S not in (3, 7) or min(desc.R, desc.S) != 1 or desc.stride != 1 or desc.epilogue not in (EPI_STORE, EPI_AFFINE) \
            or desc.accumulate or desc.skip or desc.rscale != 0.0 or (desc.epilogue == EPI_AFFINE and desc.stats_partial) \
            or desc.bn_bwd_stats or desc.relu_bits:
        return False
    if desc.H_in != desc.H_out or desc.W_in != desc.W_out or not (64 <= desc.H_out * desc.W_out <= 289):
        return False
    if desc.N < int(os.environ.get("MBX_RESIDENT_MIN_IMAGES", min_images)):
        return False
    if not desc.x or not desc.w or not desc.y:          # (a bare geometry probe, e.g. the engine's planning pass)
        # the library's own table (csrc/convr.hip mbx_launch_resident): three taps on an 8 x 8 map with C_in 192 / 224 / 256,
        # seven taps on 65 .. 289 pixels with C_in 128 / 160 / 192 -- so that a planning pass never answers "resident" for a
        # shape the launch then refuses (the 14 x 14 block8 maps of the 512 x 512 configuration: ADVICE round 5)
        hw = desc.H_out * desc.W_out
        if desc.C_out % 8:
            return False
        if desc.R * desc.S == 3:
            return hw == 64 and desc.C_in in (192, 224, 256)
        return 65 <= hw <= 289 and desc.C_in in (128, 160, 192)
    keep = desc.tile_config
    desc.tile_config = RESIDENT_TILE_CONFIG
    ok = _lib.lib().mbx_conv_supported(C.byref(desc)) == 0
    desc.tile_config = keep
    return ok


PWRES_DEFAULT = "2"                            # data gradients only: -0.035 ms per step (LAB_NOTES round 6); "1" adds the residual forwards (level or slower)
PWRES_TILE_CONFIG = 99                         # the pixel-resident pointwise launch for the epilogue-bound 1x1 layers (csrc/convr.hip, conv_pwres_kernel)


def pwres_applies(desc: ConvDesc, min_pixels=4096):
    """The pixel-resident pointwise launch by rule: a 1x1 / stride-1 convolution with C_in 96 / 128 / 320 / 384 / 448 whose epilogue
    streams trunk tensors -- the residual "up" convolutions of block35 / block17 / block8 (model.py:19-23, 39-43, 59-63) and the
    accumulate (+ relu mask) data gradients of their fused first 1x1s -- with at least 256 output channels.  MBX_PWRES=0 turns it
    off (A/B), MBX_PWRES=2 keeps it for the data gradients only.  The library has the last word (mbx_conv_supported)."""
    mode = os.environ.get("MBX_PWRES", PWRES_DEFAULT)
    if mode == "0":
        return False
    if desc.R != 1 or desc.S != 1 or desc.stride != 1 or desc.pad_t or desc.pad_l or desc.stats_partial or desc.bn_bwd_stats:
        return False
    if desc.C_in not in (96, 128, 320, 384, 448) or desc.C_out < 256 or desc.C_out % 8:
        return False
    res = desc.epilogue == EPI_RESIDUAL
    accm = desc.epilogue == EPI_STORE and desc.relu_bits and not desc.skip
    if not (res or accm) or (mode == "2" and not accm):
        return False
    return desc.N * desc.H_out * desc.W_out >= int(os.environ.get("MBX_PWRES_MIN_PIXELS", min_pixels))


def splitk_slices(desc: ConvDesc, n_cus=256):
    """Split-K slices for a forward convolution by rule (0: none): long K (>= 8192) and at most 96 tiles of 128 x 64, i.e.
    less than half the CUs busy for hundreds of K steps -- the two 3x3 head convolutions on the 1536-channel feature map
    at BATCH_SIZE 64 (80 us each at 89-135 TFLOP/s without it).  MBX_SPLITK=0 turns it off (A/B)."""
    if os.environ.get("MBX_SPLITK", "1") == "0" or desc.transposed or desc.epilogue != EPI_STORE or desc.accumulate or desc.skip:
        return 0
    ktot = desc.R * desc.S * desc.C_in
    M = desc.N * desc.H_out * desc.W_out
    tiles = ((M + 127) // 128) * ((desc.C_out + 63) // 64)
    if ktot < 8192 or tiles > 96:
        return 0
    s = int(os.environ.get("MBX_SPLITK_SLICES", "0")) or max(2, min(16, (2 * n_cus) // tiles))
    return min(s, 32)
_TUNED = {}          # repr(shape key) -> tile_config: one measurement per distinct conv in a process
TUNE_STATS = {"hits": 0, "remeasured": 0, "rejected": 0}    # table entries used as they are / shapes measured here / entries refused
_TUNE_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune_cache.json")


def _load_tune_cache():
    """Measured choices for the shapes of the shipped configurations (written on an MI355X with MBX_TUNE_SAVE=1).
    A shape that is not in the file is measured at engine build; MBX_TUNE_CACHE=0 ignores the file."""
    if os.environ.get("MBX_TUNE_CACHE", "1") != "0" and os.path.exists(_TUNE_FILE):
        try:
            import json
            with open(_TUNE_FILE) as f:
                _TUNED.update({k: int(v) for k, v in json.load(f).items()})
        except (OSError, ValueError):
            pass


def save_tune_cache(path=None):
    import json
    with open(path or _TUNE_FILE, "w") as f:
        json.dump(dict(sorted(_TUNED.items())), f, indent=0)


_load_tune_cache()


def autotune(desc: ConvDesc, key, candidates=None, iters=10):
    """Time mbx_conv(desc) for the library's own pick (0) and the given tile configurations (1-based) on the
    descriptor's real buffers; set desc.tile_config to the fastest and return it.  Results of the convolution do
    not depend on the choice.  Cached per `key` so that equal layers (and later Net instances) agree."""
    key = repr(key)
    if key in _TUNED:
        desc.tile_config = _TUNED[key]
        # make sure the table's choice still APPLIES to this shape (a table written by an older library whose kernels covered
        # other shapes must not turn into MBX_ERR_UNSUPPORTED in the middle of a training step): mbx_conv_supported runs
        # every check of mbx_conv without a launch -- nothing is written, nothing asynchronous can fail behind it
        if _lib.lib().mbx_conv_supported(C.byref(desc)) == 0:
            TUNE_STATS["hits"] += 1
            return desc.tile_config
        del _TUNED[key]                                # (also dropped from the file by the next save_tune_cache)
        TUNE_STATS["rejected"] += 1
    TUNE_STATS["remeasured"] += 1
    if candidates is None:
        candidates = (0, 2, 4, 5, 6, 9, 10, 12, 13, 14)      # the tiles that won somewhere on the B=64 layer shapes
        if os.environ.get("MBX_AUTOTUNE_SET") == "all":
            candidates = tuple(range(0, N_TILE_CONFIGS + 1))
    l = _lib.lib()
    best, best_t = 0, float("inf")
    s = _stream()
    iters = int(os.environ.get("MBX_TUNE_ITERS", iters))
    for cfg in candidates:
        desc.tile_config = cfg
        if l.mbx_conv(C.byref(desc), s) != 0:          # e.g. a configuration that does not apply
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            l.mbx_conv(C.byref(desc), s)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1)
        if t < best_t * 0.97 or (cfg == 0 and t < best_t):       # a challenger must win by 3 %
            best, best_t = cfg, t
    desc.tile_config = best
    _TUNED[key] = best
    return best


def autotune_wgrad(desc: ConvDesc, dy: View, scale, dw, db, key, candidates=(0, 2, 3, 4, 7, 8, 9, 10), iters=10):
    """The same for mbx_conv_wgrad_scaled: block shape / pixel-split count (mbx.h, tile_config 1..4).  The launches add
    into `dw`: call it where dw is scratch (the engine zeroes its gradient buffer at the start of every step)."""
    key = repr(key)
    if key in _TUNED:
        desc.tile_config = _TUNED[key]
        return desc.tile_config
    l = _lib.lib()
    s = _stream()
    best, best_t = 0, float("inf")
    iters = int(os.environ.get("MBX_TUNE_ITERS", iters))
    if os.environ.get("MBX_AUTOTUNE_WG"):
        candidates = tuple(int(v) for v in os.environ["MBX_AUTOTUNE_WG"].split(","))
    args = (dy.ptr, dy.img_stride, dy.ld, float(scale), dw.data_ptr(), _p(db), s)
    for cfg in candidates:
        desc.tile_config = cfg
        if l.mbx_conv_wgrad_scaled(C.byref(desc), *args) != 0:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            l.mbx_conv_wgrad_scaled(C.byref(desc), *args)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1)
        if t < best_t * 0.97 or (cfg == 0 and t < best_t):
            best, best_t = cfg, t
    desc.tile_config = best
    _TUNED[key] = best
    return best


class WgradGroup:
    """One grouped weight-gradient launch (mbx_conv_wgrad_grouped): plan built once from the jobs' descriptors,
    table image kept in device memory."""

    def __init__(self, jobs, deterministic=False, device="cuda"):
        l = _lib.lib()
        self.n = len(jobs)
        arr = (WgradJob * self.n)(*jobs)
        flags = (1 if deterministic else 0) | (2 if os.environ.get("MBX_WGRAD_SCATTER") == "1" else 0)   # 2: A/B knob
        nbytes = l.mbx_wgrad_plan_bytes(arr, self.n, flags)
        assert nbytes > 0
        host = (C.c_uint8 * nbytes)()
        self.info = WgradPlanInfo()
        _lib.check(l.mbx_wgrad_plan(arr, self.n, flags, host, nbytes, C.byref(self.info)), "mbx_wgrad_plan")
        import numpy as np
        self.host_image = np.frombuffer(host, dtype=np.uint8).copy()
        self.image = torch.from_numpy(self.host_image).to(device)
        self.flops = float(self.info.flops)

    def launch(self, max_workgroups=0):
        """max_workgroups > 0: a capped grid, for a launch that runs beside the next segment's backward chain on another
        stream (Trainer: overlapped weight gradients)."""
        _lib.check(_lib.lib().mbx_conv_wgrad_grouped_capped(self.image.data_ptr(), C.byref(self.info), int(max_workgroups),
                                                            _stream()), "mbx_conv_wgrad_grouped")

    def tally(self):
        """(work items processed, launches completed) since the image was uploaded -- host sync."""
        o = int(self.info.tally_off)
        t = self.image[o:o + 16].cpu().numpy().view("<u8")
        return int(t[0]), int(t[1])

    def completed_ok(self):
        """Every completed launch processed all of its work items (a launch that finds stale queue heads processes
        none and would leave dW at zero without any error)."""
        items, launches = self.tally()
        return items == launches * int(self.info.n_items)

    def reset(self):
        """Re-upload the plan image (queue heads, exit counter and tally back to zero)."""
        self.image.copy_(torch.from_numpy(self.host_image))
