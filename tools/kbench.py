"""Micro-benchmark of the conv kernels on the layer shapes of the B=64 training step.
Prints time and TFLOP/s per shape for forward, data-gradient and weight-gradient launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from multibox_amd import ops

B = int(os.environ.get("KB_B", "64"))
# name, H, W, Cin, Cout, R, S, stride, pad(t,l), count per step
SHAPES = [
    ("b17_up_1x1_384_1088", 17, 17, 384, 1088, 1, 1, 1, (0, 0), 20),
    ("b17_fused_1x1_1088_320", 17, 17, 1088, 320, 1, 1, 1, (0, 0), 20),
    ("b17_1x7_128_160", 17, 17, 128, 160, 1, 7, 1, (0, 3), 20),
    ("b17_7x1_160_192", 17, 17, 160, 192, 7, 1, 1, (3, 0), 20),
    ("b35_fused_1x1_320_96", 35, 35, 320, 96, 1, 1, 1, (0, 0), 10),
    ("b35_3x3_32_32", 35, 35, 32, 32, 3, 3, 1, (1, 1), 10),
    ("b35_3x3_48_64", 35, 35, 48, 64, 3, 3, 1, (1, 1), 10),
    ("b35_up_1x1_128_320", 35, 35, 128, 320, 1, 1, 1, (0, 0), 10),
    ("b8_fused_1x1_2080_384", 8, 8, 2080, 384, 1, 1, 1, (0, 0), 10),
    ("b8_1x3_192_224", 8, 8, 192, 224, 1, 3, 1, (0, 1), 10),
    ("b8_3x1_224_256", 8, 8, 224, 256, 3, 1, 1, (1, 0), 10),
    ("b8_up_1x1_448_2080", 8, 8, 448, 2080, 1, 1, 1, (0, 0), 10),
    ("stem_3x3_32_32_149", 149, 149, 32, 32, 3, 3, 1, (0, 0), 1),
    ("stem_3x3_32_64_147", 147, 147, 32, 64, 3, 3, 1, (1, 1), 1),
    ("b35_3x3_32_48", 35, 35, 32, 48, 3, 3, 1, (1, 1), 10),
    ("stem_3x3_80_192_73", 73, 73, 80, 192, 3, 3, 1, (0, 0), 1),
    ("m6a_3x3s2_320_384", 35, 35, 320, 384, 3, 3, 2, (0, 0), 1),
    ("m6a_3x3_256_256", 35, 35, 256, 256, 3, 3, 1, (1, 1), 1),
    ("c7b_1x1_2080_1536", 8, 8, 2080, 1536, 1, 1, 1, (0, 0), 1),
]
only = os.environ.get("KB_ONLY")
iters = int(os.environ.get("KB_ITERS", "20"))


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print("%-26s %9s %8s | %9s %8s | %9s %8s" % ("shape (B=%d)" % B, "fwd us", "TF/s", "dgrad us", "TF/s", "wgrad us", "TF/s"))
for name, H, W, Ci, Co, R, S, st, (pt, pl), cnt in SHAPES:
    if only and only not in name:
        continue
    Ho = (H + 2 * pt - R) // st + 1
    Wo = (W + 2 * pl - S) // st + 1
    # KB_LDX / KB_LDY: pixel pitch (channels) of the input / output buffers, as when the tensors are channel slices of a
    # wider buffer (block17's 672-channel branch buffer: KB_LDX=672); default = dense
    ldx, ldy = int(os.environ.get("KB_LDX", "0")) or Ci, int(os.environ.get("KB_LDY", "0")) or Co
    ldx, ldy = max(ldx, Ci), max(ldy, Co)
    x = ops.View.alloc(B, H, W, ldx).slice(0, Ci); x.buf.normal_()
    y = ops.View.alloc(B, Ho, Wo, ldy).slice(0, Co)
    dy = ops.View.alloc(B, Ho, Wo, ldy).slice(0, Co); dy.buf.normal_()
    dx = ops.View.alloc(B, H, W, ldx).slice(0, Ci)
    w = (torch.randn(Co, R, S, Ci, device="cuda") * 0.05).to(torch.bfloat16)
    wT = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
    dw = torch.zeros((Co, R, S, Ci), dtype=torch.float32, device="cuda")
    flops = 2.0 * B * Ho * Wo * Co * R * S * Ci
    kb_cfg = int(os.environ.get("KB_CFG", "0"))      # mbx_conv_desc.tile_config of the forward / data-gradient launches
    import ctypes as _C
    from multibox_amd import _lib as _L
    d_f = ops.make_desc(x, w, Co, R, S, st, pt, pl, y)
    d_f.tile_config = kb_cfg
    if _L.lib().mbx_conv_supported(_C.byref(d_f)) != 0:     # (a configuration that does not apply: the statistics rows of the default)
        d_f.tile_config = 0
    rows = ops.conv_stats_rows(d_f)
    stats = torch.zeros((rows, Co, 2), device="cuda")
    d_f = ops.make_desc(x, w, Co, R, S, st, pt, pl, y, stats=stats)
    d_d = ops.make_desc(dy, wT, Ci, R, S, st, R - 1 - pt, S - 1 - pl, dx, transposed=1)
    d_w = ops.make_desc(x, None, Co, R, S, st, pt, pl, y)
    if os.environ.get("KB_EPI") == "res":             # the epilogue-heavy forms of the residual stages:
        skip = ops.View.alloc(B, Ho, Wo, Co); skip.buf.normal_()      # forward: relu(skip + s * (acc + bias))
        bias = torch.zeros(Co, device="cuda")
        d_f = ops.make_desc(x, w, Co, R, S, st, pt, pl, y, epilogue=ops.EPI_RESIDUAL, relu=1, shift=bias, skip=skip, rscale=0.1)
        act = ops.View.alloc(B, H, W, Ci); act.buf.normal_()         # data gradient: accumulate + relu mask
        # KB_NOMASK=1: the same launch without the mask read (what a packed mask could save at most)
        d_d = ops.make_desc(dy, wT, Ci, R, S, st, R - 1 - pt, S - 1 - pl, dx, transposed=1, accumulate=1,
                            skip=None if os.environ.get("KB_NOMASK") else act)
    d_f.tile_config = kb_cfg; d_d.tile_config = kb_cfg
    import ctypes as _C
    from multibox_amd import _lib as _L
    for d_ in (d_f, d_d):                              # a configuration that does not apply to this shape: library default
        if _L.lib().mbx_conv_supported(_C.byref(d_)) != 0:
            d_.tile_config = 0
    tf = timeit(lambda: ops.conv(d_f))
    td = timeit(lambda: ops.conv(d_d))
    tw = timeit(lambda: ops.conv_wgrad(d_w, dy, dw)) if not os.environ.get("KB_NO_WGRAD") else 1.0
    tot["fwd"] += tf * cnt; tot["dgrad"] += td * cnt; tot["wgrad"] += tw * cnt
    print("%-26s %9.1f %8.1f | %9.1f %8.1f | %9.1f %8.1f  cfg f%d d%d" % (name, tf, flops / tf / 1e6, td, flops / td / 1e6, tw, flops / tw / 1e6, d_f.tile_config, d_d.tile_config))
print("weighted per-step totals (ms): fwd %.2f dgrad %.2f wgrad %.2f" % (tot["fwd"] / 1e3, tot["dgrad"] / 1e3, tot["wgrad"] / 1e3))
