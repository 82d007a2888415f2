"""What the fused launches of round 6 cost against the launches they replace, per layer shape of the B=64 training step, and
where the time of a tail goes (MBX_FUSED_PROBE bits: 16 no barrier wait, 32 no totals / statistics loads, 64 no sweep, 128 no
sums -- wrong results, timing only).  Every figure = one captured hipGraph of REP back-to-back repetitions / REP.

  forward:  [conv + stats] + [bn_apply_fused_mapped]      vs  [conv + stats + apply tail]
  backward: [dgrad] + [bn_bwd_onepass per layer]          vs  [dgrad + BN-backward tail]
"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from multibox_amd import ops, _lib

l = _lib.lib()
B = int(os.environ.get("KB_B", "64"))
REP = 20
# name, H, W, Cin, Cout, R, S, pad(t,l), fwd tile_config, dgrad tile_config, split of Cin into BN layers (backward)
SHAPES = [
    ("b17_fused_1x1_1088_320", 17, 17, 1088, 320, 1, 1, (0, 0), 36, None, None),
    ("b17_1x7_128_160", 17, 17, 128, 160, 1, 7, (0, 3), 98, 98, (128,)),
    ("b17_7x1_160_192", 17, 17, 160, 192, 7, 1, (3, 0), 98, 98, (160,)),
    ("b17_up_384_1088", 17, 17, 384, 1088, 1, 1, (0, 0), None, 36, (192, 192)),
    ("b35_fused_1x1_320_96", 35, 35, 320, 96, 1, 1, (0, 0), 35, None, None),
    ("b35_3x3_48_64", 35, 35, 48, 64, 3, 3, (1, 1), 97, 97, (48,)),
    ("b35_up_128_320", 35, 35, 128, 320, 1, 1, (0, 0), None, 35, (32, 32, 64)),
    ("b8_fused_1x1_2080_384", 8, 8, 2080, 384, 1, 1, (0, 0), 34, None, None),
    ("b8_1x3_192_224", 8, 8, 192, 224, 1, 3, (0, 1), 98, 98, (192,)),
    ("b8_up_448_2080", 8, 8, 448, 2080, 1, 1, (0, 0), None, 33, (192, 256)),
]
only = os.environ.get("KB_ONLY")


def graph_time(fn, clear):
    clear(); fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(REP):
            clear()
            fn()
    for _ in range(2):
        gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        gr.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 / REP * 1e3


def bar_block():
    bar = torch.zeros(ops.GRID_BARRIER_BYTES // 4 + 32, dtype=torch.int32, device="cuda")
    return bar, bar.data_ptr() + 4 * ((-(bar.data_ptr() // 4)) % 32)


print("probe bits: %s   batch %d   (us per launch group; 'clear' = the fill of the rows / barrier block both forms need, timed alone and subtracted)"
      % (os.environ.get("MBX_FUSED_PROBE", "0"), B))
for name, H, W, Ci, Co, R, S, (pt, pl), cf, cd, split in SHAPES:
    if only and only not in name:
        continue
    M = B * H * W
    x = ops.View.alloc(B, H, W, Ci); x.buf.normal_()
    w = (torch.randn(Co, R, S, Ci, device="cuda") * 0.05).to(torch.bfloat16)
    if cf is not None:
        y = ops.View.alloc(B, H, W, Co)
        a = ops.View.alloc(B, H, W, Co)
        table = torch.zeros((8, Co, 2), dtype=torch.int64, device="cuda")
        bar, bptr = bar_block()
        beta = torch.zeros(Co, device="cuda"); mean = torch.zeros(Co, device="cuda"); rstd = torch.zeros(Co, device="cuda"); thr = torch.zeros(Co, device="cuda")
        var = torch.zeros(Co, device="cuda")
        d = ops.make_desc(x, w, Co, R, S, 1, pt, pl, y, stats=table, stats_rows_mod=8, stats_ld=Co)
        d.tile_config = cf
        fd = ops.ConvDesc.from_buffer_copy(d)
        ba = ops.BnApplyDesc()
        ba.barrier, ba.a, ba.ld_a, ba.beta, ba.mean, ba.rstd = bptr, a.ptr, a.ld, beta.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        ba.moving_var, ba.relu_thr, ba.relu, ba.eps, ba.decay = var.data_ptr(), thr.data_ptr(), 1, 0.001, -1.0
        fd.bn_apply = C.addressof(ba)
        assert l.mbx_conv_supported(C.byref(d)) == 0 and l.mbx_conv_supported(C.byref(fd)) == 0, name

        def clear():
            table.zero_(); bar.zero_()

        def split_f():
            l.mbx_conv(C.byref(d), torch.cuda.current_stream().cuda_stream)
            l.mbx_bn_apply_fused_mapped(table.data_ptr(), 8, M, 0.001, -1.0, y.ptr, M, Co, beta.data_ptr(), 1, a.ptr, a.ld, None,
                                        mean.data_ptr(), rstd.data_ptr(), None, var.data_ptr(), thr.data_ptr(), torch.cuda.current_stream().cuda_stream)
        t_clear = graph_time(lambda: None, clear)
        t_conv = graph_time(lambda: l.mbx_conv(C.byref(d), torch.cuda.current_stream().cuda_stream), clear) - t_clear
        t_split = graph_time(split_f, clear) - t_clear
        t_fused = graph_time(lambda: l.mbx_conv(C.byref(fd), torch.cuda.current_stream().cuda_stream), clear) - t_clear
        print("fwd  %-26s conv %6.1f  conv + apply launch %6.1f  FUSED %6.1f   (tail %5.1f vs apply launch %5.1f)"
              % (name, t_conv, t_split, t_fused, t_fused - t_conv, t_split - t_conv))
    if cd is not None:
        dyX = ops.View.alloc(B, H, W, Co); dyX.buf.normal_()
        wT = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
        da = ops.View.alloc(B, H, W, Ci)
        d = ops.make_desc(dyX, wT, Ci, R, S, 1, R - 1 - pt, S - 1 - pl, da, transposed=1)
        d.tile_config = cd
        layers = []
        for K in split:
            yy = (torch.randn(M, K, device="cuda") * 1.5).to(torch.bfloat16)
            layers.append(dict(K=K, y=yy, mean=yy.float().mean(0).contiguous(), rstd=(1 / torch.sqrt(yy.float().var(0) + 0.001)).contiguous(),
                               beta=torch.zeros(K, device="cuda"), dy=torch.zeros((M, K), dtype=torch.bfloat16, device="cuda"),
                               dbeta=torch.zeros(K, device="cuda"), acc=torch.zeros((ops.BN_BWD_SLOTS, 2, K), device="cuda"),
                               ws=torch.zeros(l.mbx_bn_bwd_onepass_workspace_bytes(K) // 4, device="cuda")))
        bar, bptr = bar_block()
        t = ops.BnBwdFused()
        t.barrier, t.n = bptr, len(layers)
        c0 = 0
        for i, L in enumerate(layers):
            t.c_begin[i] = c0
            t.y[i], t.ld_y[i], t.dy[i], t.ld_dy[i] = L["y"].data_ptr(), L["K"], L["dy"].data_ptr(), L["K"]
            t.mean[i], t.rstd[i], t.beta[i], t.dbeta[i] = L["mean"].data_ptr(), L["rstd"].data_ptr(), L["beta"].data_ptr(), L["dbeta"].data_ptr()
            t.acc[i], t.acc_ld[i], t.relu[i] = L["acc"].data_ptr(), L["K"], 1
            c0 += L["K"]
        fd = ops.ConvDesc.from_buffer_copy(d)
        fd.bn_bwd = C.addressof(t)
        assert l.mbx_conv_supported(C.byref(d)) == 0 and l.mbx_conv_supported(C.byref(fd)) == 0, name

        def clear():
            bar.zero_()
            for L in layers:
                L["acc"].zero_(); L["ws"].zero_()

        def split_b():
            l.mbx_conv(C.byref(d), torch.cuda.current_stream().cuda_stream)
            c0 = 0
            for L in layers:
                dav = da.slice(c0, L["K"])
                l.mbx_bn_bwd_onepass(dav.ptr, dav.ld, 1, L["y"].data_ptr(), M, L["K"], L["mean"].data_ptr(), L["rstd"].data_ptr(), L["beta"].data_ptr(),
                                     L["dbeta"].data_ptr(), L["dy"].data_ptr(), L["ws"].data_ptr(), 0, None, torch.cuda.current_stream().cuda_stream)
                c0 += L["K"]
        t_clear = graph_time(lambda: None, clear)
        t_conv = graph_time(lambda: l.mbx_conv(C.byref(d), torch.cuda.current_stream().cuda_stream), clear) - t_clear
        t_split = graph_time(split_b, clear) - t_clear
        t_fused = graph_time(lambda: l.mbx_conv(C.byref(fd), torch.cuda.current_stream().cuda_stream), clear) - t_clear
        print("bwd  %-26s dgrad %5.1f  dgrad + %d BN-bwd launches %6.1f  FUSED %6.1f   (tail %5.1f vs BN-bwd launches %5.1f)"
              % (name, t_conv, len(layers), t_split, t_fused, t_fused - t_conv, t_split - t_conv))
