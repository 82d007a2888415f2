"""Where the time of a resident-image convolution launch (tile_config 98, csrc/convr.hip) goes: image + first tap landed,
K loop, epilogue -- from wall_clock64() stamps the kernel writes in a DEBUG build (-DMBX_I5_STAMPS; this tool rebuilds libmbx
with it on the box it runs on).
usage: MBX_BUILD_DEFS=-DMBX_I5_STAMPS python tools/res_stamps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

assert "MBX_I5_STAMPS" in os.environ.get("MBX_BUILD_DEFS", ""), "run with MBX_BUILD_DEFS=-DMBX_I5_STAMPS"
buf = torch.zeros(64 * 8 * 4, dtype=torch.int64, device="cuda")
os.environ["MBX_I5_STAMP_PTR"] = str(buf.data_ptr())
from multibox_amd import build as B_
B_.build(force=True)
from multibox_amd import ops

B = int(os.environ.get("KB_B", "64"))


def run(name, H, W, Ci, Co, R, S, pt, pl, epi, dgrad):
    if dgrad:
        x = ops.View.alloc(B, H, W, Co); y = ops.View.alloc(B, H, W, Ci)
        w = (torch.randn(Ci, R, S, Co, device="cuda") * 0.05).to(torch.bfloat16)
        x.buf.normal_()
        d = ops.make_desc(x, w, Ci, R, S, 1, R - 1 - pt, S - 1 - pl, y, transposed=1)
    else:
        x = ops.View.alloc(B, H, W, Ci); y = ops.View.alloc(B, H, W, Co)
        x.buf.normal_()
        w = (torch.randn(Co, R, S, Ci, device="cuda") * 0.05).to(torch.bfloat16)
        kw = {}
        if epi == "stats":
            kw = dict(stats=torch.zeros((B, Co, 2), device="cuda"))
        elif epi == "stats8":
            kw = dict(stats=torch.zeros((8, Co, 2), dtype=torch.int64, device="cuda"), stats_rows_mod=8, stats_ld=Co)
        elif epi == "affine":
            kw = dict(epilogue=ops.EPI_AFFINE, relu=1, scale=torch.ones(Co, device="cuda"), shift=torch.zeros(Co, device="cuda"))
        d = ops.make_desc(x, w, Co, R, S, 1, pt, pl, y, **kw)
    d.tile_config = ops.RESIDENT_TILE_CONFIG
    for _ in range(3):
        ops.conv(d)
    torch.cuda.synchronize()
    buf.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.conv(d); b.record()
    torch.cuda.synchronize()
    t = buf.cpu().reshape(64, 8, 4)[:, 0, :].double() / 100.0          # us (100 MHz clock)
    t0 = float(t[:, 0].min())
    med = lambda v: float(v.median())
    print("%-22s %-7s launch %6.1f us | start +%.2f (spread %.2f)  image+tap0 %.2f  K loop %.2f  epilogue %.2f  | end of last workgroup +%.2f us" % (
        name, "dgrad" if dgrad else epi, a.elapsed_time(b) * 1e3, med(t[:, 0]) - t0, float(t[:, 0].max()) - t0, med(t[:, 1] - t[:, 0]),
        med(t[:, 2] - t[:, 1]), med(t[:, 3] - t[:, 2]), float(t[:, 3].max()) - t0), flush=True)


for epi in ("store", "stats", "stats8", "affine"):
    run("b17_1x7_128_160", 17, 17, 128, 160, 1, 7, 0, 3, epi, False)
    run("b17_7x1_160_192", 17, 17, 160, 192, 7, 1, 3, 0, epi, False)
run("b17_1x7_128_160", 17, 17, 128, 160, 1, 7, 0, 3, "", True)
run("b17_7x1_160_192", 17, 17, 160, 192, 7, 1, 3, 0, "", True)
