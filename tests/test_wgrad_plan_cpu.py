"""Host side of the grouped weight gradient (mbx_wgrad_plan, include/mbx.h): the work-item table and the per-XCD
queues are built on the host, so they are checked without a GPU -- every (layer, output tile) must be covered by
pixel ranges that partition [0, M) exactly once, the queues must partition the item table, the deterministic flag
must forbid every split, and a panel group (all tiles of one layer and pixel range) must sit in ONE queue."""
import ctypes as C

import numpy as np
import pytest
import torch

from multibox_amd import _lib, ops


class _Item(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("layer", "tile_n", "tile_k", "m_begin", "m_end", "single", "cfg", "p2")]


# tile shapes of the grouped launch, in 64-channel / 64-column sub-images (kWgCfgs in csrc/conv.hip)
WG_CFGS = [(1, 4), (1, 5), (2, 2), (2, 3), (2, 4), (3, 2), (3, 3)]


def _jobs():
    """Layer shapes of one backward segment at BATCH_SIZE 64 (block17 x 2, a block35 3x3, a stem layer)."""
    B = 64
    shapes = [  # H, W, Cin, Cout, R, S, pad_t, pad_l
        (17, 17, 384, 1088, 1, 1, 0, 0), (17, 17, 1088, 320, 1, 1, 0, 0), (17, 17, 128, 160, 1, 7, 0, 3),
        (17, 17, 160, 192, 7, 1, 3, 0), (17, 17, 384, 1088, 1, 1, 0, 0), (35, 35, 32, 48, 3, 3, 1, 1),
        (35, 35, 320, 96, 1, 1, 0, 0), (147, 147, 32, 64, 3, 3, 1, 1),
    ]
    jobs, keep = [], []
    for (H, W, Ci, Co, R, S, pt, pl) in shapes:
        j = ops.WgradJob()
        d = j.desc
        d.x, d.x_img_stride, d.ldx = 0x10000, H * W * Ci, Ci            # fake, 16-byte aligned device addresses: the plan
        d.N, d.H_in, d.W_in, d.C_in = B, H, W, Ci                        # only embeds them
        d.C_out, d.R, d.S, d.stride, d.pad_t, d.pad_l = Co, R, S, 1, pt, pl
        d.H_out, d.W_out = H, W
        j.dy, j.dy_img_stride, j.ld_dy = 0x20000, H * W * Co, Co
        j.scale, j.dw, j.db = 1.0, 0x30000, None
        jobs.append(j)
    return jobs


def _plan(jobs, flags):
    l = _lib.lib()
    arr = (ops.WgradJob * len(jobs))(*jobs)
    nbytes = l.mbx_wgrad_plan_bytes(arr, len(jobs), flags)
    assert nbytes > 0
    host = (C.c_uint8 * nbytes)()
    info = ops.WgradPlanInfo()
    assert l.mbx_wgrad_plan(arr, len(jobs), flags, host, nbytes, C.byref(info)) == 0
    raw = bytes(host)
    items = (_Item * info.n_items).from_buffer_copy(raw[info.items_off:info.items_off + 32 * info.n_items])
    q = np.frombuffer(raw[info.queues_off:info.queues_off + 64], dtype=np.int32)
    assert info.heads_off + 9 * 32 * 4 <= nbytes and info.heads_off % 128 == 0
    assert not any(raw[info.heads_off:info.heads_off + 9 * 32 * 4])          # queue heads + exit counter start at zero
    return info, items, q


@pytest.mark.parametrize("flags", [0, 1, 2])
def test_plan_covers_every_tile_once(flags):
    jobs = _jobs()
    info, items, q = _plan(jobs, flags)
    assert info.n_layers == len(jobs)
    # queues partition the item table
    begin, end = q[:8], q[8:]
    assert begin[0] == 0 and end[-1] == info.n_items and all(begin[1:] == end[:-1]) and all(end >= begin)
    cover, cfg_of = {}, {}
    for i, it in enumerate(items):
        cover.setdefault((it.layer, it.tile_n, it.tile_k), []).append((it.m_begin, it.m_end, i))
        assert 0 <= it.cfg < len(WG_CFGS) and cfg_of.setdefault(it.layer, it.cfg) == it.cfg     # one shape per layer
    flops = 0.0
    for j, job in enumerate(jobs):
        d = job.desc
        M, Kt = d.N * d.H_out * d.W_out, d.R * d.S * d.C_in
        flops += 2.0 * M * d.C_out * Kt
        ny, nx = WG_CFGS[cfg_of[j]]
        tn, tk = -(-d.C_out // (64 * ny)), -(-Kt // (64 * nx))
        for a in range(tn):
            for b in range(tk):
                r = sorted(cover.pop((j, a, b)))
                assert r[0][0] == 0 and r[-1][1] == M, (j, a, b, r[:2])
                assert all(x[1] == y[0] for x, y in zip(r, r[1:])), "pixel ranges must tile [0, M) without gap or overlap"
                assert all(x[0] % 64 == 0 for x in r)
                if flags & 1:
                    assert len(r) == 1, "MBX_WGRAD_DETERMINISTIC: one adder per dw element"
                assert all(items[x[2]].single == (1 if len(r) == 1 else 0) for x in r)     # plain stores only when alone
    assert not cover, "items outside every layer's tile grid"
    assert abs(info.flops - flops) <= 1e-9 * flops
    if not flags & 2:
        # a panel group (layer, pixel range) is dealt to ONE queue: its tiles share dy / x rows through that XCD's L2
        qof = lambda i: int(np.searchsorted(end, i, side="right"))
        groups = {}
        for i, it in enumerate(items):
            groups.setdefault((it.layer, it.m_begin), set()).add(qof(i))
        assert all(len(v) == 1 for v in groups.values())
    if flags == 0:
        # and the queues carry comparable work (64-pixel steps); un-split tiles (flag 1) cannot balance, blocks steal
        qof = lambda i: int(np.searchsorted(end, i, side="right"))
        load = np.zeros(8)
        for i, it in enumerate(items):
            load[qof(i)] += (it.m_end - it.m_begin + 63) // 64
        assert load.min() > 0 and load.max() <= 1.35 * load.mean(), load
