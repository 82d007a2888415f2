"""Micro-benchmark of the GROUPED weight-gradient launch (mbx_conv_wgrad_grouped) one layer shape at a time, and of
all of them in one launch: time, useful TFLOP/s, and the padded work the tiles really do (tile-steps x tile area).
usage: python tools/wgbench.py            (KB_B=64 batch; KB_ONLY=substring filter)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from multibox_amd import ops

B = int(os.environ.get("KB_B", "64"))
SHAPES = [  # name, H, W, Cin, Cout, R, S, stride, pad(t,l), count per step
    ("b17_up_1x1_384_1088", 17, 17, 384, 1088, 1, 1, 1, (0, 0), 20),
    ("b17_fused_1x1_1088_320", 17, 17, 1088, 320, 1, 1, 1, (0, 0), 20),
    ("b17_1x7_128_160", 17, 17, 128, 160, 1, 7, 1, (0, 3), 20),
    ("b17_7x1_160_192", 17, 17, 160, 192, 7, 1, 1, (3, 0), 20),
    ("b35_fused_1x1_320_96", 35, 35, 320, 96, 1, 1, 1, (0, 0), 10),
    ("b35_3x3_32_32", 35, 35, 32, 32, 3, 3, 1, (1, 1), 10),
    ("b35_3x3_32_48", 35, 35, 32, 48, 3, 3, 1, (1, 1), 10),
    ("b35_3x3_48_64", 35, 35, 48, 64, 3, 3, 1, (1, 1), 10),
    ("b35_up_1x1_128_320", 35, 35, 128, 320, 1, 1, 1, (0, 0), 10),
    ("b8_fused_1x1_2080_384", 8, 8, 2080, 384, 1, 1, 1, (0, 0), 10),
    ("b8_1x3_192_224", 8, 8, 192, 224, 1, 3, 1, (0, 1), 10),
    ("b8_3x1_224_256", 8, 8, 224, 256, 3, 1, 1, (1, 0), 10),
    ("b8_up_1x1_448_2080", 8, 8, 448, 2080, 1, 1, 1, (0, 0), 10),
    ("stem_3x3_8_32_299s2", 299, 299, 8, 32, 3, 3, 2, (0, 0), 1),
    ("stem_3x3_32_32_149", 149, 149, 32, 32, 3, 3, 1, (0, 0), 1),
    ("stem_3x3_32_64_147", 147, 147, 32, 64, 3, 3, 1, (1, 1), 1),
    ("stem_1x1_64_80_73", 73, 73, 64, 80, 1, 1, 1, (0, 0), 1),
    ("stem_3x3_80_192_73", 73, 73, 80, 192, 3, 3, 1, (0, 0), 1),
    ("m6a_3x3s2_320_384", 35, 35, 320, 384, 3, 3, 2, (0, 0), 1),
    ("m6a_3x3_256_256", 35, 35, 256, 256, 3, 3, 1, (1, 1), 1),
    ("m7a_fused_1x1_1088_768", 17, 17, 1088, 768, 1, 1, 1, (0, 0), 1),
    ("c7b_1x1_2080_1536", 8, 8, 2080, 1536, 1, 1, 1, (0, 0), 1),
]
only = os.environ.get("KB_ONLY")
iters = int(os.environ.get("KB_ITERS", "10"))


class Item(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("layer", "tile_n", "tile_k", "m_begin", "m_end", "single", "cfg", "p2")]


WG = [(1, 4), (1, 5), (2, 2), (2, 3), (2, 4), (3, 2), (3, 3)]      # kWgCfgs (csrc/conv.hip)


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def make_job(H, W, Ci, Co, R, S, st, pt, pl):
    Ho = (H + 2 * pt - R) // st + 1
    Wo = (W + 2 * pl - S) // st + 1
    x = ops.View.alloc(B, H, W, Ci); x.buf.normal_()
    dy = ops.View.alloc(B, Ho, Wo, Co); dy.buf.normal_()
    y = ops.View.alloc(1, Ho, Wo, 8)
    dw = torch.zeros((Co, R, S, Ci), dtype=torch.float32, device="cuda")
    j = ops.WgradJob()
    j.desc = ops.make_desc(x, None, Co, R, S, st, pt, pl, ops.View(y.buf, 1, Ho, Wo, 8))
    j.desc.H_out, j.desc.W_out = Ho, Wo
    j.dy, j.dy_img_stride, j.ld_dy = dy.ptr, dy.img_stride, dy.ld
    j.scale, j.dw, j.db = 1.0, dw.data_ptr(), None
    flops = 2.0 * B * Ho * Wo * Co * R * S * Ci
    return j, flops, (x, dy, dw)


def padded_flops(group, jobs):
    raw = group.host_image.tobytes()
    info = group.info
    items = (Item * info.n_items).from_buffer_copy(raw[info.items_off:info.items_off + 32 * info.n_items])
    tot = 0.0
    for it in items:
        steps = (it.m_end - it.m_begin + 63) // 64
        ny, nx = WG[it.cfg]; tot += steps * 2.0 * 64 * (64 * ny) * (64 * nx)
    return tot, info.n_items


print("%-26s %9s %8s %8s %7s" % ("shape (B=%d)" % B, "us", "TF/s", "padTF/s", "items"))
alljobs, keep, tot_us, tot_fl = [], [], 0.0, 0.0
for name, H, W, Ci, Co, R, S, st, (pt, pl), cnt in SHAPES:
    if only and only not in name:
        continue
    j, fl, bufs = make_job(H, W, Ci, Co, R, S, st, pt, pl)
    keep.append(bufs)
    grp = ops.WgradGroup([j])
    t = timeit(grp.launch)
    pf, n = padded_flops(grp, [j])
    print("%-26s %9.1f %8.1f %8.1f %7d" % (name, t, fl / t / 1e6, pf / t / 1e6, n))
    for _ in range(min(cnt, 4)):
        alljobs.append(j)
    tot_us += t * cnt
    tot_fl += fl * cnt
print("sum over a step (count-weighted, one launch per layer): %.2f ms, %.0f TF/s" % (tot_us / 1e3, tot_fl / tot_us / 1e6))
grp = ops.WgradGroup(alljobs)
t = timeit(grp.launch)
fl = grp.flops
pf, n = padded_flops(grp, alljobs)
print("one grouped launch of %d jobs: %.1f us, %.0f TF/s useful, %.0f TF/s padded, %d items" % (len(alljobs), t, fl / t / 1e6, pf / t / 1e6, n))
