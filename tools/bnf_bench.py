"""Times the forward batch-norm step after a conv (statistics finalize + normalise/ReLU) on the network's layer shapes:
the two-launch form (bn_finalize + bn_apply) against the fused launch (every workgroup re-reduces the partial rows of
its 64 channels).  usage: MBX_BN_FUSED_ROWS=.. MBX_BN_FUSED_BLOCKS_MANY=.. python tools/bnf_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from multibox_amd import _lib
l = _lib.lib()
S = lambda: torch.cuda.current_stream().cuda_stream
# (M, C, tile height of the conv that produced the partial rows, launches per step)
shapes = [(78400, 96, 128, 10), (78400, 32, 128, 10), (78400, 64, 128, 10), (78400, 48, 128, 10), (18496, 320, 128, 20),
          (18496, 160, 128, 20), (18496, 192, 128, 20), (4096, 384, 64, 10), (4096, 224, 64, 10), (4096, 256, 64, 10),
          (4096, 1536, 128, 1), (78400, 256, 128, 2), (18496, 768, 128, 1), (322624, 192, 256, 1), (1382976, 64, 256, 1)]
tot = 0.0
for M, C, BM, cnt in shapes:
    y = (torch.randn(M, C) * 2).to(torch.bfloat16).cuda()
    a = torch.empty_like(y)
    rows = (M + BM - 1) // BM
    part = torch.rand((rows, C, 2), device="cuda")
    mean, rstd, beta = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    mm, mv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")

    def fused():
        _lib.check(l.mbx_bn_apply_fused(part.data_ptr(), rows, M, 0.001, 0.9997, y.data_ptr(), M, C, beta.data_ptr(), 1, a.data_ptr(), C,
                                        mean.data_ptr(), rstd.data_ptr(), mm.data_ptr(), mv.data_ptr(), S()))
    for _ in range(3):
        fused()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fused()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    tot += t * cnt
    print("M=%7d C=%4d rows=%5d  %7.1f us  (%.2f TB/s @4B)" % (M, C, rows, t, M * C * 4e-6 / t))
print("count-weighted total: %.3f ms" % (tot / 1e3))
