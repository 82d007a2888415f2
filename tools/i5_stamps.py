"""Where a tile's time goes inside the persistent igemm5 kernel: K loop, epilogue rows, statistics + hand-over, from
wall_clock64() stamps the kernel writes in a DEBUG build (-DMBX_I5_STAMPS; this tool rebuilds libmbx with it on the
box it runs on -- rebuild without MBX_BUILD_DEFS afterwards, or take a fresh checkout).
usage: MBX_BUILD_DEFS=-DMBX_I5_STAMPS python tools/i5_stamps.py [cfg ...]     (tile_config 33..37)"""
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
TILES = {33: "128x64", 34: "128x128", 35: "192x128", 36: "256x128", 37: "256x64", 38: "128x192", 39: "128x256"}


def run(name, H, W, Ci, Co, R, S, pt, pl, cfg, epi):
    x = ops.View.alloc(B, H, W, Ci); x.buf.normal_()
    y = ops.View.alloc(B, H, W, Co)
    w = (torch.randn(Co, R, S, Ci, device="cuda") * 0.05).to(torch.bfloat16)
    kw = {}
    if epi == "res":
        skip = ops.View.alloc(B, H, W, Co); skip.buf.normal_()
        kw = dict(epilogue=ops.EPI_RESIDUAL, relu=1, shift=torch.zeros(Co, device="cuda"), skip=skip, rscale=0.1)
    elif epi == "stats":
        d0 = ops.make_desc(x, w, Co, R, S, 1, pt, pl, y); d0.tile_config = cfg
        kw = dict(stats=torch.zeros((ops.conv_stats_rows(d0), Co, 2), device="cuda"))
    d = ops.make_desc(x, w, Co, R, S, 1, pt, pl, y, **kw)
    d.tile_config = cfg
    for _ in range(3):
        ops.conv(d)
    torch.cuda.synchronize()
    buf.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.conv(d); b.record()
    torch.cuda.synchronize()
    t = buf.cpu().reshape(64, 8, 4).double() / 100.0          # us (100 MHz clock)
    blk = t[8]
    n = int((blk[:, 0] > 0).sum())
    med = lambda v: float(v.median()) if len(v) else float("nan")
    nk = (R * S * Ci + 63) // 64
    kl, ep = blk[:n, 1] - blk[:n, 0], blk[:n, 2] - blk[:n, 1]
    raw = buf.cpu().reshape(64, 8, 4)[8].double()
    cyc = raw[:n, 3]                                           # core-clock cycles of the K loop (+ the epilogue's pre-issued reads)
    ghz = float((cyc / ((raw[:n, 1] - raw[:n, 0]) * 10.0)).median()) if n else float("nan")   # cycles per nanosecond
    gap = blk[1:n, 0] - blk[:n - 1, 2]
    print("%-24s %-7s %-5s launch %6.1f us | tiles/block %d  K loop %.2f us (%d steps: %.2f us each)  rows %.2f us  stats+next %.2f us  | core clock in the K loop %.2f GHz" % (
        name, TILES[cfg], epi, a.elapsed_time(b) * 1e3, n, med(kl), nk, med(kl) / nk, med(ep), med(gap), ghz), flush=True)


cfgs = [int(a) for a in sys.argv[1:]] or [33, 34, 35, 36, 37]
for cfg in cfgs:
    for epi in ("store", "stats", "res"):
        run("b17_up_1x1_384_1088", 17, 17, 384, 1088, 1, 1, 0, 0, cfg, epi)
    if os.environ.get("I5S_UP_ONLY"):
        continue
    run("b17_fused_1x1_1088_320", 17, 17, 1088, 320, 1, 1, 0, 0, cfg, "stats")
    run("b17_1x7_128_160", 17, 17, 128, 160, 1, 7, 0, 3, cfg, "stats")
    run("m6a_3x3_256_256", 35, 35, 256, 256, 3, 3, 1, 1, cfg, "stats")
