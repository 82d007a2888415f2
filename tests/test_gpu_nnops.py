"""GPU parity of the HBM-bound kernels (batch norm, pooling, glue, optimizer) vs torch-CPU float32.

Tolerance: bf16 outputs within 1 bf16 ulp of the float32 reference computed from the same
bf16-rounded inputs (2^-7 relative + 2e-3 of max|ref|); float32 outputs rtol 1e-5.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available()
    return torch


def bfr(torch, t):
    return t.to(torch.bfloat16).to(torch.float32)


def close_bf16(out, ref):
    out, ref = out.float().cpu(), ref.float().cpu()
    mx = float(ref.abs().max()) + 1e-20
    bad = (out - ref).abs() > (2.0 ** -7) * ref.abs() + 2e-3 * mx
    return int(bad.sum()) == 0, "%d/%d bad, max err %.3g, max|ref| %.3g" % (int(bad.sum()), bad.numel(), float((out - ref).abs().max()), mx)


def S():
    import torch
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("M,Cc,relu", [(2 * 35 * 35, 96, 1), (3 * 17 * 17, 320, 1), (64, 1536, 0), (5000, 32, 1)])
def test_bn_forward_backward(T, M, Cc, relu):
    torch = T
    from multibox_amd import _lib, ops
    l = _lib.lib()
    gen = torch.Generator().manual_seed(M + Cc)
    y = bfr(torch, torch.randn(M, Cc, generator=gen) * 2 + 0.5)
    beta = torch.randn(Cc, generator=gen) * 0.3
    da = bfr(torch, torch.randn(M, Cc, generator=gen))
    # reference (train.py:94-99 semantics: no gamma, eps .001, biased variance)
    yr = y.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    mean, var = yr.mean(0), yr.var(0, unbiased=False)
    a_ref = (yr - mean) * torch.rsqrt(var + 0.001) + br
    if relu:
        a_ref = torch.relu(a_ref)
    a_q = bfr(torch, a_ref.detach())
    # stats via partial sums exactly as the conv epilogue would write them (2 partial rows)
    h = M // 2
    part = torch.stack([torch.stack([y[:h].sum(0), (y[:h] ** 2).sum(0)], 1), torch.stack([y[h:].sum(0), (y[h:] ** 2).sum(0)], 1)]).cuda().contiguous()
    dm, dr = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    mm, mv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    _lib.check(l.mbx_bn_finalize(part.data_ptr(), 2, Cc, M, 0.001, 0.9, dm.data_ptr(), dr.data_ptr(), mm.data_ptr(), mv.data_ptr(), S()))
    assert torch.allclose(dm.cpu(), mean.detach(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(dr.cpu(), torch.rsqrt(var.detach() + 0.001), rtol=1e-4)
    assert torch.allclose(mm.cpu(), 0.1 * mean.detach(), rtol=1e-4, atol=1e-6)           # moving -= (1-d)(moving - batch)
    assert torch.allclose(mv.cpu(), 0.9 + 0.1 * var.detach(), rtol=1e-4)
    yd = y.to(torch.bfloat16).cuda()
    av = ops.View.alloc(1, 1, M, Cc + 8, zero=True).slice(8, Cc)
    bd = beta.cuda()
    _lib.check(l.mbx_bn_apply(yd.data_ptr(), M, Cc, dm.data_ptr(), dr.data_ptr(), bd.data_ptr(), relu, av.ptr, av.ld, S()))
    ok, msg = close_bf16(av.tensor().reshape(M, Cc), a_ref.detach())
    assert ok, "bn_apply: " + msg
    # backward
    a_ref.backward(da)
    dav = ops.View.alloc(1, 1, M, Cc + 16, zero=True).slice(16, Cc)
    dav.tensor().reshape(M, Cc).copy_(da.to(torch.bfloat16))
    rows = l.mbx_bn_bwd_rows(M, Cc)
    partial = torch.zeros((rows, Cc, 2), device="cuda")
    dbeta = torch.ones(Cc, device="cuda")                   # accumulates on top of existing content
    m12 = torch.zeros(2 * Cc, device="cuda")
    dy = torch.zeros((M, Cc), dtype=torch.bfloat16, device="cuda")
    # two forms of the relu mask: read from the stored activation, or (a = NULL) recomputed from y and beta
    for a_ptr in (av.ptr, None):
        dbeta.fill_(1.0)
        _lib.check(l.mbx_bn_bwd_reduce(dav.ptr, dav.ld, a_ptr, av.ld, relu, yd.data_ptr(), M, Cc, dm.data_ptr(), dr.data_ptr(),
                                       bd.data_ptr(), partial.data_ptr(), S()))
        _lib.check(l.mbx_bn_bwd_finalize(partial.data_ptr(), rows, Cc, M, dbeta.data_ptr(), m12.data_ptr(), S()))
        _lib.check(l.mbx_bn_bwd_apply(dav.ptr, dav.ld, a_ptr, av.ld, relu, yd.data_ptr(), M, Cc, dm.data_ptr(), dr.data_ptr(),
                                      bd.data_ptr(), m12.data_ptr(), dy.data_ptr(), S()))
        # the kernel masks with the STORED activation (bf16 a > 0); the reference with the float one: identical sets
        # except where rounding flips a tiny positive to 0 -- tolerate through the tolerance on dbeta / dy.
        assert torch.allclose(dbeta.cpu() - 1.0, br.grad, rtol=2e-3, atol=2e-2 * float(br.grad.abs().max()))
        ok, msg = close_bf16(dy, yr.grad)
        assert ok or float((dy.float().cpu() - yr.grad).abs().max()) < 2e-2 * float(yr.grad.abs().max()), "bn_bwd: " + msg
    # ---- the fused one-launch form gives the same results as the two-launch form
    av2 = ops.View.alloc(1, 1, M, Cc + 8, zero=True).slice(8, Cc)
    dm2, dr2 = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    mm2, mv2 = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    _lib.check(l.mbx_bn_apply_fused(part.data_ptr(), 2, M, 0.001, 0.9, yd.data_ptr(), M, Cc, bd.data_ptr(), relu, av2.ptr, av2.ld,
                                    dm2.data_ptr(), dr2.data_ptr(), mm2.data_ptr(), mv2.data_ptr(), S()))
    assert torch.equal(av2.tensor(), av.tensor()) and torch.equal(dm2, dm) and torch.equal(dr2, dr)
    assert torch.equal(mm2, mm) and torch.equal(mv2, mv)

@pytest.mark.parametrize("M,Cc,relu", [(2450, 96, 1), (18496, 160, 1), (18496, 320, 1), (78400, 96, 1), (18496, 512, 0),
                                       (18496, 768, 1), (78400, 208, 1), (78400, 256, 1), (4096, 1536, 1), (1001, 24, 1),
                                       (7, 8, 0)])
def test_bn_backward_onepass(T, M, Cc, relu):
    """One-launch backward (slice in registers across a grid barrier) == the three-launch form, every register
    variant (2..20 vectors per lane), ragged rows, channel counts that do not divide the workgroup."""
    torch = T
    from multibox_amd import _lib, ops
    l = _lib.lib()
    assert l.mbx_bn_bwd_onepass_supported(M, Cc, 0) == 1
    gen = torch.Generator().manual_seed(M * 31 + Cc)
    y = (torch.randn(M, Cc, generator=gen) * 2 + 0.5).to(torch.bfloat16).cuda()
    dav = ops.View.alloc(1, 1, M, Cc + 16, zero=True).slice(8, Cc)
    dav.tensor().reshape(M, Cc).copy_(torch.randn(M, Cc, generator=gen).to(torch.bfloat16))
    mean = y.float().mean(0).contiguous()
    rstd = torch.rsqrt(y.float().var(0, unbiased=False) + 0.001).contiguous()
    beta = (torch.randn(Cc, generator=gen) * 0.3).cuda()
    rows = l.mbx_bn_bwd_rows(M, Cc)
    partial = torch.zeros((rows, Cc, 2), device="cuda")
    m12 = torch.zeros(2 * Cc, device="cuda")
    dbeta_ref, dbeta = torch.ones(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    dy_ref = torch.zeros((M, Cc), dtype=torch.bfloat16, device="cuda")
    dy = torch.full((M + 1, Cc), 7.0, dtype=torch.bfloat16, device="cuda")      # guard row: nothing written past M
    _lib.check(l.mbx_bn_bwd_reduce(dav.ptr, dav.ld, None, 0, relu, y.data_ptr(), M, Cc, mean.data_ptr(), rstd.data_ptr(),
                                   beta.data_ptr(), partial.data_ptr(), S()))
    _lib.check(l.mbx_bn_bwd_finalize(partial.data_ptr(), rows, Cc, M, dbeta_ref.data_ptr(), m12.data_ptr(), S()))
    _lib.check(l.mbx_bn_bwd_apply(dav.ptr, dav.ld, None, 0, relu, y.data_ptr(), M, Cc, mean.data_ptr(), rstd.data_ptr(),
                                  beta.data_ptr(), m12.data_ptr(), dy_ref.data_ptr(), S()))
    nws = l.mbx_bn_bwd_onepass_workspace_bytes(Cc) // 4
    for max_wg in (0, 192):     # all CUs / a capped grid (the data-parallel setting)
        if not l.mbx_bn_bwd_onepass_supported(M, Cc, max_wg):
            continue
        ws = torch.zeros(nws, device="cuda")
        dbeta.fill_(1.0)
        _lib.check(l.mbx_bn_bwd_onepass(dav.ptr, dav.ld, relu, y.data_ptr(), M, Cc, mean.data_ptr(), rstd.data_ptr(),
                                        beta.data_ptr(), dbeta.data_ptr(), dy.data_ptr(), ws.data_ptr(), max_wg, None, S()))
        torch.cuda.synchronize()
        flags = ws[8 * 2 * Cc:8 * 2 * Cc + 2].view(torch.int32).tolist()
        assert flags[1] == 0, "grid barrier timed out"
        assert flags[0] > 0
        assert torch.allclose(dbeta, dbeta_ref, rtol=1e-4, atol=1e-3 * float(dbeta_ref.abs().max()))
        ok, msg = close_bf16(dy[:M], dy_ref)
        assert ok, msg
        assert bool((dy[M] == 7.0).all())


def test_bn_backward_onepass_unsupported(T):
    from multibox_amd import _lib
    l = _lib.lib()
    assert l.mbx_bn_bwd_onepass_supported(64 * 147 * 147, 32, 0) == 0     # stem layers: three launches
    assert l.mbx_bn_bwd_onepass_supported(100, 12, 0) == 0                # channels not a multiple of 8
    assert l.mbx_bn_bwd_onepass_supported(78400, 256, 0) == 1 and l.mbx_bn_bwd_onepass_supported(78400, 256, 128) == 0


def test_bn_fold(T):
    torch = T
    from multibox_amd import _lib
    l = _lib.lib()
    Cc = 100
    mm, mv, beta = torch.randn(Cc), torch.rand(Cc) + 0.1, torch.randn(Cc)
    sc, sh = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    dmm, dmv, dbeta = mm.cuda(), mv.cuda(), beta.cuda()      # keep the device tensors alive across the call
    _lib.check(l.mbx_bn_fold(dmm.data_ptr(), dmv.data_ptr(), dbeta.data_ptr(), 0.001, Cc, sc.data_ptr(), sh.data_ptr(), S()))
    s = torch.rsqrt(mv + 0.001)
    assert torch.allclose(sc.cpu(), s, rtol=1e-5) and torch.allclose(sh.cpu(), beta - mm * s, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("N,H,W,Cc,k,st", [(2, 35, 35, 64, 3, 2), (1, 17, 17, 1088, 3, 2), (3, 9, 9, 8, 3, 2),
                                            (2, 12, 13, 16, 2, 2), (1, 11, 11, 24, 3, 1)])   # last two: run-time window
def test_maxpool(T, N, H, W, Cc, k, st):
    torch = T
    import torch.nn.functional as F
    from multibox_amd import _lib, ops
    l = _lib.lib()
    gen = torch.Generator().manual_seed(H)
    x = bfr(torch, torch.relu(torch.randn(N, H, W, Cc, generator=gen)))      # many exact ties at 0
    Ho, Wo = (H - k) // st + 1, (W - k) // st + 1
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr, idx = F.max_pool2d(xr, k, st, return_indices=True)
    xv = ops.View.alloc(N, H, W, Cc + 8, zero=True).slice(0, Cc)
    xv.tensor().copy_(x.to(torch.bfloat16))
    yv = ops.View.alloc(N, Ho, Wo, Cc + 8, zero=True).slice(8, Cc)
    am = torch.zeros((N, Ho, Wo, Cc), dtype=torch.uint8, device="cuda")
    _lib.check(l.mbx_maxpool_fwd(xv.ptr, xv.img_stride, xv.ld, N, H, W, Cc, k, st, yv.ptr, yv.img_stride, yv.ld, Ho, Wo, am.data_ptr(), S()))
    assert torch.equal(yv.tensor().float().cpu(), yr.detach().permute(0, 2, 3, 1))
    dy = bfr(torch, torch.randn(N, Ho, Wo, Cc, generator=gen))
    yr.backward(dy.permute(0, 3, 1, 2))
    dyv = ops.View.alloc(N, Ho, Wo, Cc)
    dyv.tensor().copy_(dy.to(torch.bfloat16))
    dxv = ops.View.alloc(N, H, W, Cc, zero=True)
    _lib.check(l.mbx_maxpool_bwd(dyv.ptr, dyv.img_stride, dyv.ld, am.data_ptr(), N, H, W, Cc, k, st, Ho, Wo, dxv.ptr, dxv.img_stride, dxv.ld, 0, S()))
    ok, msg = close_bf16(dxv.tensor(), xr.grad.permute(0, 2, 3, 1))      # first-maximum routing, like torch/TF
    assert ok, msg
    _lib.check(l.mbx_maxpool_bwd(dyv.ptr, dyv.img_stride, dyv.ld, am.data_ptr(), N, H, W, Cc, k, st, Ho, Wo, dxv.ptr, dxv.img_stride, dxv.ld, 1, S()))
    ok, msg = close_bf16(dxv.tensor(), 2 * xr.grad.permute(0, 2, 3, 1))
    assert ok, "accumulate: " + msg


@pytest.mark.parametrize("N,H,W,Cc,k,pad", [(2, 35, 35, 192, 3, 1), (4, 8, 8, 1536, 8, 0), (1, 14, 14, 64, 8, 0)])
def test_avgpool(T, N, H, W, Cc, k, pad):
    torch = T
    import torch.nn.functional as F
    from multibox_amd import _lib, ops
    l = _lib.lib()
    gen = torch.Generator().manual_seed(k)
    x = bfr(torch, torch.randn(N, H, W, Cc, generator=gen))
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = F.avg_pool2d(xr, k, 1, pad, count_include_pad=False)              # TF SAME: divide by valid taps
    xv = ops.View.alloc(N, H, W, Cc)
    xv.tensor().copy_(x.to(torch.bfloat16))
    yv = ops.View.alloc(N, Ho, Wo, Cc)
    _lib.check(l.mbx_avgpool_fwd(xv.ptr, xv.img_stride, xv.ld, N, H, W, Cc, k, pad, yv.ptr, yv.img_stride, yv.ld, Ho, Wo, S()))
    ok, msg = close_bf16(yv.tensor(), yr.detach().permute(0, 2, 3, 1))
    assert ok, msg
    dy = bfr(torch, torch.randn(N, Ho, Wo, Cc, generator=gen))
    yr.backward(dy.permute(0, 3, 1, 2))
    dyv = ops.View.alloc(N, Ho, Wo, Cc)
    dyv.tensor().copy_(dy.to(torch.bfloat16))
    dxv = ops.View.alloc(N, H, W, Cc, zero=True)
    _lib.check(l.mbx_avgpool_bwd(dyv.ptr, dyv.img_stride, dyv.ld, N, H, W, Cc, k, pad, Ho, Wo, dxv.ptr, dxv.img_stride, dxv.ld, 0, S()))
    ok, msg = close_bf16(dxv.tensor(), xr.grad.permute(0, 2, 3, 1))
    assert ok, msg


def test_glue_kernels(T):
    torch = T
    from multibox_amd import _lib, ops
    l = _lib.lib()
    gen = torch.Generator().manual_seed(1)
    # relu mask
    M, Cc = 777, 40
    g = bfr(torch, torch.randn(M, Cc, generator=gen))
    a = bfr(torch, torch.relu(torch.randn(M, Cc, generator=gen)))
    gv = ops.View.alloc(1, 1, M, Cc + 8, zero=True).slice(8, Cc)
    gv.tensor().reshape(M, Cc).copy_(g.to(torch.bfloat16))
    av = ops.View.alloc(1, 1, M, Cc)
    av.tensor().reshape(M, Cc).copy_(a.to(torch.bfloat16))
    _lib.check(l.mbx_relu_mask(gv.ptr, gv.ld, av.ptr, av.ld, M, Cc, S()))
    assert torch.equal(gv.tensor().reshape(M, Cc).float().cpu(), g * (a > 0))
    # pack input
    img = torch.rand(2, 5, 7, 3, generator=gen) * 2 - 1
    out = torch.zeros((2, 5, 7, 8), dtype=torch.bfloat16, device="cuda")
    dimg = img.cuda()
    _lib.check(l.mbx_pack_input(dimg.data_ptr(), 2 * 5 * 7, out.data_ptr(), S()))
    assert torch.equal(out[..., :3].float().cpu(), bfr(torch, img)) and float(out[..., 3:].float().abs().max()) == 0
    # head gather / scatter (model.py:295-322 flatten order)
    N, g_, k, P, off = 3, 6, 5, 646, 320
    cells = g_ * g_
    h = torch.randn(N * cells, 32, generator=gen)
    locs, logits = torch.zeros((N, P, 4), device="cuda"), torch.zeros((N, P), device="cuda")
    dh = h.cuda()
    _lib.check(l.mbx_head_gather(dh.data_ptr(), 32, N, cells, k, P, off, locs.data_ptr(), logits.data_ptr(), S()))
    ref_l = h[:, :4 * k].reshape(N, cells * k, 4)
    ref_c = h[:, 4 * k:5 * k].reshape(N, cells * k)
    assert torch.equal(locs[:, off:off + cells * k].cpu(), ref_l) and torch.equal(logits[:, off:off + cells * k].cpu(), ref_c)
    assert float(locs[:, :off].abs().max()) == 0
    dl, dz = torch.randn(N, P, 4, generator=gen), torch.randn(N, P, generator=gen)
    gb = torch.ones((N * cells, 32), dtype=torch.bfloat16, device="cuda")
    ddl, ddz = dl.cuda(), dz.cuda()
    _lib.check(l.mbx_head_scatter(ddl.data_ptr(), ddz.data_ptr(), N, cells, k, P, off, gb.data_ptr(), 32, S()))
    gb = gb.float().cpu()
    assert torch.equal(gb[:, :4 * k], bfr(torch, dl[:, off:off + cells * k].reshape(N * cells, 4 * k)))
    assert torch.equal(gb[:, 4 * k:5 * k], bfr(torch, dz[:, off:off + cells * k].reshape(N * cells, k)))
    assert float(gb[:, 5 * k:].abs().max()) == 0
    # every head in ONE launch (mbx_head_gather_all / mbx_head_scatter_all) = the per-head calls, bit for bit:
    # the six heads of the 299x299 / k = 5 network (grids 8, 6, 4, 3, 2 with k boxes per cell, 1 with one: P = 646)
    import ctypes as C
    N, P, ld = 3, 646, 32
    grids = [(8, 5), (6, 5), (4, 5), (3, 5), (2, 5), (1, 1)]
    heads = (_lib.Head * len(grids))()
    hs, gs, keep, off = [], [], [], 0
    for j, (g_, k) in enumerate(grids):
        hj = torch.randn(N * g_ * g_, ld, generator=gen).cuda()
        gj = torch.ones((N * g_ * g_, ld), dtype=torch.bfloat16, device="cuda")
        hs.append(hj); gs.append(gj)
        heads[j].h, heads[j].ld_h, heads[j].g, heads[j].ld_g = hj.data_ptr(), ld, gj.data_ptr(), ld
        heads[j].cells, heads[j].k, heads[j].off = g_ * g_, k, off
        off += g_ * g_ * k
    assert off == P
    la, za = torch.zeros((N, P, 4), device="cuda"), torch.zeros((N, P), device="cuda")
    lb, zb = torch.zeros((N, P, 4), device="cuda"), torch.zeros((N, P), device="cuda")
    _lib.check(l.mbx_head_gather_all(heads, len(grids), N, P, la.data_ptr(), za.data_ptr(), S()))
    for j, (g_, k) in enumerate(grids):
        _lib.check(l.mbx_head_gather(hs[j].data_ptr(), ld, N, g_ * g_, k, P, heads[j].off, lb.data_ptr(), zb.data_ptr(), S()))
    assert torch.equal(la, lb) and torch.equal(za, zb) and float(la.abs().min()) > 0
    dl, dz = torch.randn(N, P, 4, generator=gen).cuda(), torch.randn(N, P, generator=gen).cuda()
    _lib.check(l.mbx_head_scatter_all(dl.data_ptr(), dz.data_ptr(), heads, len(grids), N, P, S()))
    for j, (g_, k) in enumerate(grids):
        ref = torch.ones((N * g_ * g_, ld), dtype=torch.bfloat16, device="cuda")
        _lib.check(l.mbx_head_scatter(dl.data_ptr(), dz.data_ptr(), N, g_ * g_, k, P, heads[j].off, ref.data_ptr(), ld, S()))
        assert torch.equal(gs[j], ref), j
    # mbx_step_begin: both buffers cleared, the control word's count folded into the running total first
    G = torch.randn(4096 + 64, generator=gen).cuda()
    ws = torch.randn(1024, generator=gen).cuda()
    tot = torch.full((), 5, dtype=torch.int64, device="cuda")
    G[136] = 7.0
    sc = torch.full((1,), 3.0, device="cuda")
    _lib.check(l.mbx_step_begin(G.data_ptr(), G.numel(), ws.data_ptr(), ws.numel(), 136, tot.data_ptr(), sc.data_ptr(), S()))
    assert float(G.abs().max()) == 0 and float(ws.abs().max()) == 0 and int(tot) == 12 and float(sc) == 0.0
    _lib.check(l.mbx_step_begin(G.data_ptr(), G.numel(), ws.data_ptr(), ws.numel(), 136, tot.data_ptr(), None, S()))
    assert int(tot) == 12


def test_filter_prepare(T):
    torch = T
    from multibox_amd import _lib
    l = _lib.lib()
    gen = torch.Generator().manual_seed(2)
    shapes = [(32, 3, 3, 8), (25, 1, 1, 96), (160, 1, 7, 128), (96, 2, 2, 128), (70, 3, 1, 40), (16, 1, 1, 12)]   # last: scalar path
    ws = [bfr(torch, torch.randn(*s, generator=gen)) for s in shapes]
    src_off, dst_off, blocks, entries, refs = 0, 0, 0, [], []
    flat = []
    for w, (K, R, S_, Cc) in zip(ws, shapes):
        kpad = (K + 7) // 8 * 8
        entries.append(_lib.FilterEntry(src_off, dst_off, K, R, S_, Cc, kpad, blocks))
        flat.append(w.reshape(-1))
        ref = torch.zeros(Cc, R, S_, kpad)
        ref[..., :K] = w.flip(1, 2).permute(3, 1, 2, 0)
        refs.append((dst_off, ref))
        n = Cc * R * S_ * kpad
        src_off += w.numel()
        dst_off += (n + 7) // 8 * 8
        blocks += R * S_ * ((Cc + 31) // 32) * ((kpad + 63) // 64)
    wsrc = torch.cat(flat).to(torch.bfloat16).cuda()
    wdst = torch.zeros(dst_off, dtype=torch.bfloat16, device="cuda")
    arr = (_lib.FilterEntry * len(entries))(*entries)
    raw = np.frombuffer(C.string_at(C.addressof(arr), C.sizeof(arr)), dtype=np.uint8).copy()
    table = torch.from_numpy(raw).cuda()
    _lib.check(l.mbx_filter_prepare(wsrc.data_ptr(), wdst.data_ptr(), table.data_ptr(), len(entries), blocks, S()))
    for off, ref in refs:
        assert torch.equal(wdst[off:off + ref.numel()].float().cpu(), ref.reshape(-1))


def test_rmsprop_ema_step(T):
    """train.py:190-263 on a flat range vs the numpy restatement (oracle/ref_numpy.py)."""
    torch = T
    from multibox_amd import _lib
    from oracle import ref_numpy as R
    l = _lib.lib()
    rng = np.random.RandomState(0)
    n = 100003
    w, g = rng.randn(n).astype(np.float32), (rng.randn(n) * 10).astype(np.float32)
    ms, ema = np.ones(n, np.float32), w.copy()
    lr, decay, eps, wd, d = 0.0094, 0.9, 1.0, 4e-5, R.ema_decay(0.9999, 3)
    dw, dg, dms, dema = [torch.from_numpy(a.copy()).cuda() for a in (w, g, ms, ema)]
    wb = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    reg = torch.zeros(1, device="cuda")
    for _ in range(2):
        _lib.check(l.mbx_rmsprop_ema_step(dw.data_ptr(), dg.data_ptr(), dms.data_ptr(), None, dema.data_ptr(), wb.data_ptr(), n,
                                          lr, decay, 0.0, eps, wd, d, 1, reg.data_ptr(), None, S()))
    reg_ref = 0.0
    for _ in range(2):
        reg_ref += 0.5 * wd * float((w.astype(np.float64) ** 2).sum())
        ema -= np.float32(1 - d) * (ema - w)
        gg = g + np.float32(wd) * w
        R.rmsprop_step(w, gg, ms, np.zeros_like(w), lr, decay, 0.0, eps)
    assert np.allclose(dw.cpu().numpy(), w, rtol=1e-5, atol=1e-6)
    assert np.allclose(dms.cpu().numpy(), ms, rtol=1e-5)
    assert np.allclose(dema.cpu().numpy(), ema, rtol=1e-5, atol=1e-6)
    assert np.isclose(float(reg), reg_ref, rtol=1e-4)
    assert torch.equal(wb.float().cpu(), dw.to(torch.bfloat16).float().cpu())
    # frozen range: no update, EMA + bf16 refresh only
    w0 = dw.clone()
    _lib.check(l.mbx_rmsprop_ema_step(dw.data_ptr(), None, None, None, dema.data_ptr(), wb.data_ptr(), n, lr, decay, 0.0, eps, wd, d, 0, None, None, S()))
    assert torch.equal(dw, w0)
    # step control block: a non-zero word (barrier timeouts | stop requests) makes the launch a no-op
    for ctl in ([1.0, 0.0], [0.0, 2.0]):
        skip = torch.tensor(ctl, device="cuda")
        before = [t.clone() for t in (dw, dms, dema, wb, reg)]
        _lib.check(l.mbx_rmsprop_ema_step(dw.data_ptr(), dg.data_ptr(), dms.data_ptr(), None, dema.data_ptr(), wb.data_ptr(), n,
                                          lr, decay, 0.0, eps, wd, d, 1, reg.data_ptr(), skip.data_ptr(), S()))
        _lib.check(l.mbx_ema_update(dema.data_ptr(), dw.data_ptr(), n, d, skip.data_ptr(), S()))
        assert all(torch.equal(a, b) for a, b in zip(before, (dw, dms, dema, wb, reg)))
    skip = torch.zeros(2, device="cuda")
    _lib.check(l.mbx_rmsprop_ema_step(dw.data_ptr(), dg.data_ptr(), dms.data_ptr(), None, dema.data_ptr(), wb.data_ptr(), n,
                                      lr, decay, 0.0, eps, wd, d, 1, reg.data_ptr(), skip.data_ptr(), S()))
    assert not torch.equal(dw, w0)


@pytest.mark.parametrize("M,Ks,offs,relu", [(2 * 35 * 35, (32, 48), (96, 192), 1), (4096, (384, 288), (0, 384), 1),
                                             (2450, (64, 96), (208, 432), 1), (777, (24, 8, 40), (64, 8, 128), 0)])
def test_bn_group_entry_points(T, M, Ks, offs, relu):
    """BATCH-NORM GROUPS (round 4, mbx.h): sibling layers normalised by ONE finalize / apply / backward launch over their
    concatenated [M, sum K] tensors, the strided activation / gradient view addressed through an mbx_chan_map, against
    the per-layer launches: finalize, apply and the three-launch backward bit-identical; the one-launch backward (fp32
    atomics) within the tolerance of test_bn_backward_onepass; channels of the view outside the members untouched."""
    torch = T
    from multibox_amd import _lib, ops
    l = _lib.lib()
    gen = torch.Generator().manual_seed(M + sum(Ks))
    Kt, n = sum(Ks), len(Ks)
    koff = [sum(Ks[:i]) for i in range(n)]
    ld = max(o + k for o, k in zip(offs, Ks)) + 16
    y = (torch.randn(M, Kt, generator=gen) * 2 + 0.5).to(torch.bfloat16).cuda()
    beta = (torch.randn(Kt, generator=gen) * 0.3).cuda()
    rows = [3, 5, 2, 4][:n]
    parts = [torch.rand(r, k, 2, generator=gen).cuda() * M for r, k in zip(rows, Ks)]
    for p_, k, ko in zip(parts, Ks, koff):              # consistent sums: s2 >= s1^2 / M
        yy = y[:, ko:ko + k].float()
        p_[:, :, 0] = yy.sum(0) / p_.shape[0]
        p_[:, :, 1] = (yy * yy).sum(0) / p_.shape[0]
    rel = [o - ko for o, ko in zip(offs, koff)]
    base = min(rel)
    cm = _lib.ChanMap()
    cm.n = n
    for i in range(n):
        cm.c_begin[i], cm.offset[i] = koff[i], rel[i] - base
    # ---- forward: group finalize + apply vs per member
    mean_g, rstd_g, mm_g, mv_g = (torch.zeros(Kt, device="cuda") for _ in range(4))
    mv_g.fill_(1.0)
    pa = (C.c_void_p * n)(*[p_.data_ptr() for p_ in parts])
    _lib.check(l.mbx_bn_finalize_parts(pa, (C.c_int32 * n)(*rows), (C.c_int32 * n)(*Ks), n, M, 0.001, 0.9, mean_g.data_ptr(),
                                       rstd_g.data_ptr(), mm_g.data_ptr(), mv_g.data_ptr(), S()))
    a_g = torch.full((M, ld), 7.0, dtype=torch.bfloat16, device="cuda")
    _lib.check(l.mbx_bn_apply_mapped(y.data_ptr(), M, Kt, mean_g.data_ptr(), rstd_g.data_ptr(), beta.data_ptr(), relu,
                                     a_g.data_ptr() + 2 * base, ld, C.byref(cm), S()))
    a_r = torch.full((M, ld), 7.0, dtype=torch.bfloat16, device="cuda")
    mean_r, rstd_r, mm_r, mv_r = (torch.zeros(Kt, device="cuda") for _ in range(4))
    mv_r.fill_(1.0)
    for p_, r, k, ko, o in zip(parts, rows, Ks, koff, offs):
        yk = y[:, ko:ko + k].contiguous()
        _lib.check(l.mbx_bn_finalize(p_.data_ptr(), r, k, M, 0.001, 0.9, mean_r.data_ptr() + 4 * ko, rstd_r.data_ptr() + 4 * ko,
                                     mm_r.data_ptr() + 4 * ko, mv_r.data_ptr() + 4 * ko, S()))
        _lib.check(l.mbx_bn_apply(yk.data_ptr(), M, k, mean_r.data_ptr() + 4 * ko, rstd_r.data_ptr() + 4 * ko, beta.data_ptr() + 4 * ko,
                                  relu, a_r.data_ptr() + 2 * o, ld, S()))
    torch.cuda.synchronize()
    assert torch.equal(mean_g, mean_r) and torch.equal(rstd_g, rstd_r) and torch.equal(mm_g, mm_r) and torch.equal(mv_g, mv_r)
    assert torch.equal(a_g, a_r)
    assert float((a_g[:, :min(offs)].float() - 7.0).abs().max() if min(offs) else 0.0) == 0
    # ---- backward: three launches (deterministic) and one launch, gradient view through the map
    da = torch.zeros((M, ld), dtype=torch.bfloat16, device="cuda")
    for k, o in zip(Ks, offs):
        da[:, o:o + k] = torch.randn(M, k, generator=gen).to(torch.bfloat16).cuda()
    rows_b = l.mbx_bn_bwd_rows(M, Kt)
    partial = torch.zeros((rows_b, Kt, 2), device="cuda")
    dbeta_g, m12 = torch.ones(Kt, device="cuda"), torch.zeros(2 * Kt, device="cuda")
    dy_g = torch.zeros((M, Kt), dtype=torch.bfloat16, device="cuda")
    _lib.check(l.mbx_bn_bwd_reduce_mapped(da.data_ptr() + 2 * base, ld, None, 0, relu, y.data_ptr(), M, Kt, mean_g.data_ptr(),
                                          rstd_g.data_ptr(), beta.data_ptr(), partial.data_ptr(), C.byref(cm), S()))
    _lib.check(l.mbx_bn_bwd_finalize(partial.data_ptr(), rows_b, Kt, M, dbeta_g.data_ptr(), m12.data_ptr(), S()))
    _lib.check(l.mbx_bn_bwd_apply_mapped(da.data_ptr() + 2 * base, ld, None, 0, relu, y.data_ptr(), M, Kt, mean_g.data_ptr(),
                                         rstd_g.data_ptr(), beta.data_ptr(), m12.data_ptr(), dy_g.data_ptr(), C.byref(cm), S()))
    dy_r, dbeta_r = torch.zeros((M, Kt), dtype=torch.bfloat16, device="cuda"), torch.ones(Kt, device="cuda")
    for k, ko, o in zip(Ks, koff, offs):
        yk, dyk = y[:, ko:ko + k].contiguous(), torch.zeros((M, k), dtype=torch.bfloat16, device="cuda")
        rk = l.mbx_bn_bwd_rows(M, k)
        pk, m12k = torch.zeros((rk, k, 2), device="cuda"), torch.zeros(2 * k, device="cuda")
        args = (da.data_ptr() + 2 * o, ld, None, 0, relu, yk.data_ptr(), M, k, mean_g.data_ptr() + 4 * ko, rstd_g.data_ptr() + 4 * ko,
                beta.data_ptr() + 4 * ko)
        _lib.check(l.mbx_bn_bwd_reduce(*args, pk.data_ptr(), S()))
        _lib.check(l.mbx_bn_bwd_finalize(pk.data_ptr(), rk, k, M, dbeta_r.data_ptr() + 4 * ko, m12k.data_ptr(), S()))
        _lib.check(l.mbx_bn_bwd_apply(*args, m12k.data_ptr(), dyk.data_ptr(), S()))
        dy_r[:, ko:ko + k] = dyk
    torch.cuda.synchronize()
    # (the partial-row grouping of the reduce differs with the channel count: sums agree to float32 rounding, dy to 1 ulp)
    assert torch.allclose(dbeta_g, dbeta_r, rtol=1e-4, atol=1e-3)
    ok, msg = close_bf16(dy_g, dy_r)
    assert ok, "group three-launch backward: " + msg
    if l.mbx_bn_bwd_onepass_supported(M, Kt, 0):
        ws = torch.zeros(l.mbx_bn_bwd_onepass_workspace_bytes(Kt) // 4, device="cuda")
        dy_o, dbeta_o = torch.zeros((M, Kt), dtype=torch.bfloat16, device="cuda"), torch.ones(Kt, device="cuda")
        _lib.check(l.mbx_bn_bwd_onepass_mapped(da.data_ptr() + 2 * base, ld, relu, y.data_ptr(), M, Kt, mean_g.data_ptr(), rstd_g.data_ptr(),
                                               beta.data_ptr(), dbeta_o.data_ptr(), dy_o.data_ptr(), ws.data_ptr(), 0, None, C.byref(cm), S()))
        torch.cuda.synchronize()
        assert int(ws.view(torch.int32)[8 * 2 * Kt + 1]) == 0                       # no barrier time-out
        assert torch.allclose(dbeta_o, dbeta_g, rtol=1e-4, atol=1e-3)
        ok, msg = close_bf16(dy_o, dy_g)
        assert ok, "group one-launch backward: " + msg


@pytest.mark.parametrize("N,H,W,Cc,relu", [(2, 21, 21, 64, 1), (3, 15, 17, 192, 1), (1, 9, 9, 8, 0)])
def test_bn_backward_pooled(T, N, H, W, Cc, relu):
    """Round 4: the three-launch BN backward of a layer that feeds only a 3x3 / 2 max-pool, gathering its activation gradient
    from the pool's output gradient on the fly (mbx_bn_bwd_reduce_pooled / _apply_pooled) == mbx_maxpool_fwd's argmax ->
    mbx_maxpool_bwd -> mbx_bn_bwd_reduce / finalize / apply: the same per-element gradients (same additions, same bf16 rounding), statistics
    summed in another grouping (2 x 2 pixel blocks share their four window loads)."""
    torch = T
    from multibox_amd import _lib
    l = _lib.lib()
    gen = torch.Generator().manual_seed(N * 1000 + H * 10 + Cc)
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    M = N * H * W
    y = (torch.randn(M, Cc, generator=gen) * 2 + 0.5).to(torch.bfloat16).cuda()
    mean = y.float().mean(0).contiguous()
    rstd = torch.rsqrt(y.float().var(0, unbiased=False) + 0.001).contiguous()
    beta = (torch.randn(Cc, generator=gen) * 0.3).cuda()
    a = torch.zeros((N, H, W, Cc), dtype=torch.bfloat16, device="cuda")
    _lib.check(l.mbx_bn_apply(y.data_ptr(), M, Cc, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), relu, a.data_ptr(), Cc, S()))
    p = torch.zeros((N, Ho, Wo, Cc), dtype=torch.bfloat16, device="cuda")
    arg = torch.zeros((N, Ho, Wo, Cc), dtype=torch.uint8, device="cuda")
    _lib.check(l.mbx_maxpool_fwd(a.data_ptr(), H * W * Cc, Cc, N, H, W, Cc, 3, 2, p.data_ptr(), Ho * Wo * Cc, Cc, Ho, Wo, arg.data_ptr(), S()))
    # forward twin: normalise + pool in one pass (the activation never stored) == the two launches, bit for bit
    p2 = torch.zeros((N, Ho, Wo, Cc + 8), dtype=torch.bfloat16, device="cuda")
    arg2 = torch.zeros((N, Ho, Wo, Cc), dtype=torch.uint8, device="cuda")
    _lib.check(l.mbx_bn_apply_maxpool(y.data_ptr(), N, H, W, Cc, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), relu,
                                      p2.data_ptr() + 16, Ho * Wo * (Cc + 8), Cc + 8, Ho, Wo, arg2.data_ptr(), S()))
    torch.cuda.synchronize()
    assert torch.equal(p2[..., 8:], p) and torch.equal(arg2, arg) and float(p2[..., :8].float().abs().max()) == 0
    gy = torch.randn(N, Ho, Wo, Cc, generator=gen).to(torch.bfloat16).cuda()
    # reference path: pool backward -> stored da -> reduce / finalize / apply
    da = torch.zeros((N, H, W, Cc), dtype=torch.bfloat16, device="cuda")
    _lib.check(l.mbx_maxpool_bwd(gy.data_ptr(), Ho * Wo * Cc, Cc, arg.data_ptr(), N, H, W, Cc, 3, 2, Ho, Wo, da.data_ptr(), H * W * Cc, Cc, 0, S()))
    out = []
    for pooled in (False, True):
        rows = l.mbx_bn_bwd_rows_pooled(N, H, W, Cc) if pooled else l.mbx_bn_bwd_rows(M, Cc)
        partial = torch.zeros((rows, Cc, 2), device="cuda")
        dbeta, m12 = torch.zeros(Cc, device="cuda"), torch.zeros(2 * Cc, device="cuda")
        dy = torch.zeros((M, Cc), dtype=torch.bfloat16, device="cuda")
        stat = (mean.data_ptr(), rstd.data_ptr(), beta.data_ptr())
        if pooled:
            geo = (gy.data_ptr(), Ho * Wo * Cc, Cc, arg.data_ptr(), N, H, W, Ho, Wo, relu, y.data_ptr(), Cc) + stat
            _lib.check(l.mbx_bn_bwd_reduce_pooled(*geo, partial.data_ptr(), S()))
            _lib.check(l.mbx_bn_bwd_finalize(partial.data_ptr(), rows, Cc, M, dbeta.data_ptr(), m12.data_ptr(), S()))
            _lib.check(l.mbx_bn_bwd_apply_pooled(*geo, m12.data_ptr(), dy.data_ptr(), S()))
        else:
            args = (da.data_ptr(), Cc, None, 0, relu, y.data_ptr(), M, Cc) + stat
            _lib.check(l.mbx_bn_bwd_reduce(*args, partial.data_ptr(), S()))
            _lib.check(l.mbx_bn_bwd_finalize(partial.data_ptr(), rows, Cc, M, dbeta.data_ptr(), m12.data_ptr(), S()))
            _lib.check(l.mbx_bn_bwd_apply(*args, m12.data_ptr(), dy.data_ptr(), S()))
        torch.cuda.synchronize()
        out.append((partial, dbeta, m12, dy))
    # per-element gradients are the stored ones; the partial sums are grouped by 2 x 2 pixel blocks instead of rows:
    # totals equal to float32 rounding, dy to 1 bf16 ulp
    assert torch.allclose(out[0][0].double().sum(0), out[1][0].double().sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(out[0][1], out[1][1], rtol=1e-5, atol=1e-3) and torch.allclose(out[0][2], out[1][2], rtol=1e-5, atol=1e-6)
    ok, msg = close_bf16(out[1][3], out[0][3])
    assert ok, msg
    assert float(out[0][3].float().abs().max()) > 0
