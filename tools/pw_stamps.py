"""Where the time of a pixel-resident pointwise launch (tile_config 99, csrc/convr.hip conv_pwres_kernel) goes, per column tile:
K loop, epilogue -- from wall_clock64() stamps of a DEBUG build (-DMBX_I5_STAMPS; this tool rebuilds libmbx with it).
usage: MBX_BUILD_DEFS=-DMBX_I5_STAMPS python tools/pw_stamps.py"""
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


def run(name, H, W, Ci, Co, kind):
    x = ops.View.alloc(B, H, W, Ci); x.buf.normal_()
    y = ops.View.alloc(B, H, W, Co)
    w = (torch.randn(Co, 1, 1, Ci, device="cuda") * 0.05).to(torch.bfloat16)
    skip = ops.View.alloc(B, H, W, Co); skip.buf.normal_()
    M = B * H * W
    bits = torch.zeros((M, (Co + 31) // 32 * 4), dtype=torch.uint8, device="cuda")
    if kind == "res":
        d = ops.make_desc(x, w, Co, 1, 1, 1, 0, 0, y, epilogue=ops.EPI_RESIDUAL, relu=1, shift=torch.zeros(Co, device="cuda"), skip=skip, rscale=0.1, relu_bits=bits)
    elif kind == "acc":
        d = ops.make_desc(x, w, Co, 1, 1, 1, 0, 0, y, accumulate=1, acc_src=skip, relu_bits=bits)
    d.tile_config = ops.PWRES_TILE_CONFIG
    for _ in range(3):
        ops.conv(d)
    torch.cuda.synchronize()
    buf.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.conv(d); b.record()
    torch.cuda.synchronize()
    t = buf.cpu().reshape(64, 8, 4).double() / 100.0          # us
    t0 = float(t[:, 0, 0].min())
    med = lambda v: float(v.median())
    line = "%-22s %-4s launch %6.1f us | first tile starts +%.2f |" % (name, kind, a.elapsed_time(b) * 1e3, med(t[:, 0, 0]) - t0)
    last = 0
    for c in range(8):
        if float(t[:, c, 0].max()) <= 0:
            break
        last = c
        kus = med(t[:, c, 1] - t[:, c, 0])
        line += " [K +%.2f %.2f epi %.2f clk %.2f GHz]" % (med(t[:, c, 0]) - t0, kus, med(t[:, c, 2] - t[:, c, 1]), med(t[:, c, 3]) * 100.0 / 1e3 / max(kus, 1e-3))
    print(line, flush=True)


if os.environ.get("PW_PROBE"):
    import ctypes
    for dbg in (0, 12, 16, 28):
        os.environ["MBX_I5_DBG"] = str(dbg)
        print("MBX_I5_DBG", dbg, "(4: no fragment reads, 8: no MFMAs, 16: no epilogue memory traffic)")
        run("b17_up_384_1088", 17, 17, 384, 1088, "res")
    sys.exit(0)
run("b17_up_384_1088", 17, 17, 384, 1088, "res")
run("b17_dg_320_1088", 17, 17, 320, 1088, "acc")
run("b35_up_128_320", 35, 35, 128, 320, "res")
run("b8_up_448_2080", 8, 8, 448, 2080, "res")
