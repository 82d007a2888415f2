"""Times the batch-norm backward forms (three launches vs one launch) on the network's layer shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from multibox_amd import _lib, ops
l = _lib.lib()
S = lambda: torch.cuda.current_stream().cuda_stream
shapes = [(78400, 96), (78400, 32), (78400, 64), (18496, 320), (18496, 160), (18496, 192), (4096, 384), (4096, 224), (4096, 1536),
          (78400, 256), (18496, 768)]
for M, C in shapes:
    y = (torch.randn(M, C) * 2).to(torch.bfloat16).cuda()
    da = ops.View.alloc(1, 1, M, C, zero=False)
    da.tensor().normal_()
    mean, rstd, beta = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    rows = l.mbx_bn_bwd_rows(M, C)
    partial = torch.zeros((rows, C, 2), device="cuda")
    m12 = torch.zeros(2 * C, device="cuda"); dbeta = torch.zeros(C, device="cuda")
    dy = torch.zeros((M, C), dtype=torch.bfloat16, device="cuda")
    ws = torch.zeros(l.mbx_bn_bwd_onepass_workspace_bytes(C) // 4, device="cuda")

    def three():
        _lib.check(l.mbx_bn_bwd_reduce(da.ptr, da.ld, None, 0, 1, y.data_ptr(), M, C, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), partial.data_ptr(), S()))
        _lib.check(l.mbx_bn_bwd_finalize(partial.data_ptr(), rows, C, M, dbeta.data_ptr(), m12.data_ptr(), S()))
        _lib.check(l.mbx_bn_bwd_apply(da.ptr, da.ld, None, 0, 1, y.data_ptr(), M, C, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), m12.data_ptr(), dy.data_ptr(), S()))

    def one():
        ws.zero_()
        _lib.check(l.mbx_bn_bwd_onepass(da.ptr, da.ld, 1, y.data_ptr(), M, C, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), dbeta.data_ptr(), dy.data_ptr(), ws.data_ptr(), 0, None, S()))

    res = []
    for f in (three, one):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    gb = M * C * 2e-9
    print("M=%6d C=%4d  three %7.1f us (%.2f TB/s @10B)  one %7.1f us (%.2f TB/s @6B)  flag=%s" % (
        M, C, res[0], gb * 5 / res[0] * 1e6 / 1e3, res[1], gb * 3 / res[1] * 1e6 / 1e3, ws[8 * C:8 * C + 2].view(torch.int32).tolist()))
