"""Do an independent data-gradient and weight-gradient launch of the same layer overlap when issued on two streams?
Compares sequential (one stream) with concurrent (two streams) time per pair, for tile configurations whose LDS
footprints can / cannot co-reside on a CU.  env: MBX_WGRAD_NG=1 -> 4-wave 64 KB weight-gradient blocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from multibox_amd import ops, _lib
import ctypes as C
l = _lib.lib()
B = 64
SHAPES = [("b17_7x1_160_192", 17, 17, 160, 192, 7, 1, (3, 0)), ("b17_up_1x1_384_1088", 17, 17, 384, 1088, 1, 1, (0, 0)),
          ("b35_3x3_48_64", 35, 35, 48, 64, 3, 3, (1, 1)), ("b17_fused_1x1_1088_320", 17, 17, 1088, 320, 1, 1, (0, 0))]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, H, W, Ci, Co, R, S, (pt, pl) in SHAPES:
    x = ops.View.alloc(B, H, W, Ci); x.buf.normal_()
    dy = ops.View.alloc(B, H, W, Co); dy.buf.normal_()
    dx = ops.View.alloc(B, H, W, Ci)
    w = (torch.randn(Co, R, S, Ci, device="cuda") * 0.05).to(torch.bfloat16)
    wT = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
    dw = torch.zeros((Co, R, S, Ci), dtype=torch.float32, device="cuda")
    d_d = ops.make_desc(dy, wT, Ci, R, S, 1, R - 1 - pt, S - 1 - pl, dx, transposed=1)
    d_w = ops.make_desc(x, None, Co, R, S, 1, pt, pl, dy)
    def dgrad(st): _lib.check(l.mbx_conv(C.byref(d_d), st.cuda_stream))
    def wgrad(st): _lib.check(l.mbx_conv_wgrad(C.byref(d_w), dy.ptr, dy.img_stride, dy.ld, dw.data_ptr(), None, st.cuda_stream))
    def run(concurrent, n=30):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        for _ in range(n):
            if concurrent:
                ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
                dgrad(s1); wgrad(s2)
                ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
            else:
                dgrad(s1); wgrad(s1)
        e1.record(s1); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    run(False, 3); run(True, 3)
    a, b = run(False), run(True)
    print("%-26s sequential %6.1f us/pair   two streams %6.1f us/pair" % (name, a, b))
