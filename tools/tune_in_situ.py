"""Re-measure the tile configuration of every conv launch INSIDE a training step and rewrite multibox_amd/tune_cache.json.

ops.autotune times a launch in isolation (the same launch repeated: its operands sit in L2 / Infinity Cache).  In the
step the operands were just written by the previous kernel, other tensors have passed through the caches since they
were last touched, and the choice that wins in isolation is not always the one that wins there.  This tool runs the
eager step (forward + loss + backward, Trainer.run_eager_once) once per candidate configuration with HIP events around
every launch -- the GPU parked behind a spin kernel while the host queues the step, as in bench.py -- and keeps, per
distinct layer shape (the cache key), the configuration with the smallest summed time.

usage: python tools/tune_in_situ.py [--batch 64] [--input-size 299] [--k 5] [--fine-tune | --infer] [--repeats 5] [--all] [--out FILE]
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONV_CANDIDATES = (2, 4, 5, 6, 9, 10, 12, 13, 14, 33, 34, 35, 36, 37, 38, 65)  # 33..38: the persistent igemm5 tiles; 65: igemm7 (panel-resident 1x1)
WGRAD_CANDIDATES = (2, 3, 4, 7, 8, 9, 10)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--input-size", type=int, default=299)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--max-num-bboxes", type=int, default=13)
    ap.add_argument("--fine-tune", action="store_true")
    ap.add_argument("--infer", action="store_true", help="the detect path: inference-mode forward only (BATCH_SIZE patches)")
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--threshold", type=float, default=0.02, help="a challenger must beat the current choice by this fraction")
    ap.add_argument("--all", action="store_true", help="try every tile configuration, not only the ones that won before")
    ap.add_argument("--out", default=None, help="cache file to write (default: the package's tune_cache.json)")
    args = ap.parse_args()
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    from multibox_amd import _lib, ops, priors as PR
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS

    if args.infer:
        net = Net(batch=args.batch, input_size=args.input_size, k=args.k, mode="infer")
        net.fold_bn()
        net.set_input(torch.rand(args.batch, args.input_size, args.input_size, 3, device="cuda") * 2 - 1)
        run_step = net.forward
    else:
        net = Net(batch=args.batch, input_size=args.input_size, k=args.k, mode="train", fine_tune=args.fine_tune, seed=2)
        pri = PR.priors_for_input_size(DEFAULT_ASPECT_RATIOS[args.k], args.input_size).astype(np.float32)
        tr = Trainer(net, pri, max_num_bboxes=args.max_num_bboxes, location_loss_alpha=1000.0, use_graph=False)
        images, gt, n = synthetic_batch(args.batch, args.input_size, args.max_num_bboxes, seed=0)
        tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())
        for _ in range(2):
            tr.step()
        run_step = tr.run_eager_once
    run_step()
    torch.cuda.synchronize()

    conv_cands = tuple(range(1, ops.N_TILE_CONFIGS + 1)) if args.all else CONV_CANDIDATES
    wgrad_cands = tuple(range(1, 11)) if args.all else WGRAD_CANDIDATES
    l = _lib.lib()
    by_addr = {C.addressof(d): (key, d, what) for key, d, what in net.tune_registry}
    tuned = {C.addressof(d): d.tile_config for _, d, _ in net.tune_registry}
    orig_conv, orig_wg, orig_bn = l.mbx_conv, l.mbx_conv_wgrad_scaled, l.mbx_bn_apply_fused
    state = {"cand": None, "kind": None, "rows": None, "recs": None}

    def timed(call, key, cfg):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = call()
        b.record()
        state["recs"].append((key, cfg, a, b))
        return r

    def conv(desc_ref, stream):
        d = desc_ref._obj
        ent = by_addr.get(C.addressof(d))
        if ent is None:
            return orig_conv(desc_ref, stream)
        if state["kind"] == "conv" and state["cand"] is not None:
            d.tile_config = state["cand"]
        if d.stats_partial:
            state["rows"] = ops.conv_stats_rows(d)            # the partial-row count follows the tile
        r = timed(lambda: orig_conv(desc_ref, stream), ent[0], d.tile_config)
        d.tile_config = tuned[C.addressof(d)]
        if r != 0:                                             # the candidate does not apply to this launch: not a time
            state["recs"].pop()
            if d.stats_partial:
                state["rows"] = ops.conv_stats_rows(d)
            r = orig_conv(desc_ref, stream)
        return r

    def wgrad(desc_ref, *rest):
        d = desc_ref._obj
        ent = by_addr.get(C.addressof(d))
        if ent is None:
            return orig_wg(desc_ref, *rest)
        if state["kind"] == "wgrad" and state["cand"] is not None:
            d.tile_config = state["cand"]
        r = timed(lambda: orig_wg(desc_ref, *rest), ent[0], d.tile_config)
        d.tile_config = tuned[C.addressof(d)]
        return r

    def bn_apply_fused(stats, rows, *rest):
        return orig_bn(stats, state["rows"] if state["rows"] is not None else rows, *rest)

    times = {}                                                # key -> cfg -> [summed ms per pass]

    def one_pass(kind, cand):
        state.update(cand=cand, kind=kind, rows=None, recs=[])
        torch.cuda.synchronize()
        torch.cuda._sleep(int(100e-3 * 2.0e9))
        run_step()
        torch.cuda.synchronize()
        acc = {}
        for key, cfg, a, b in state["recs"]:
            acc[(key, cfg)] = acc.get((key, cfg), 0.0) + a.elapsed_time(b)
        for (key, cfg), t in acc.items():
            times.setdefault(key, {}).setdefault(cfg, []).append(t)

    l.mbx_conv, l.mbx_conv_wgrad_scaled, l.mbx_bn_apply_fused = conv, wgrad, bn_apply_fused
    try:
        for rep in range(args.repeats):
            one_pass(None, None)                              # the current choices (every launch)
            for c in conv_cands:
                one_pass("conv", c)
            for c in wgrad_cands:
                one_pass("wgrad", c)
    finally:
        l.mbx_conv, l.mbx_conv_wgrad_scaled, l.mbx_bn_apply_fused = orig_conv, orig_wg, orig_bn

    what_of = {key: what for key, _, what in net.tune_registry}
    cur_of = {key: d.tile_config for key, d, _ in net.tune_registry}
    med = lambda v: sorted(v)[len(v) // 2]
    changed, gain = 0, 0.0
    for key, per_cfg in times.items():
        cur = cur_of[key]
        allowed = wgrad_cands if what_of[key] == "wgrad" else conv_cands
        # passes of the OTHER kind also ran this launch with its current choice: they all count for `cur`
        t_cur = med(per_cfg[cur])
        best, t_best = cur, t_cur
        for cfg, v in per_cfg.items():
            if cfg == cur or cfg not in allowed:
                continue
            t = med(v)
            if t < t_best and t < t_cur * (1.0 - args.threshold):
                best, t_best = cfg, t
        if best != cur:
            changed += 1
            gain += t_cur - t_best
            print("%-6s %-110s %2d -> %2d  %8.1f -> %8.1f us" % (what_of[key], key[:110], cur, best, 1e3 * t_cur, 1e3 * t_best))
            ops._TUNED[key] = best
        if best > ops.I5_FLAG:
            # data-parallel runs do not use the persistent igemm5 launch (engine._tune): keep the best igemm3 tile beside it
            i3 = {cfg: med(v) for cfg, v in per_cfg.items() if cfg <= ops.N_TILE_CONFIGS and (cfg in allowed or cfg == cur)}
            if i3:
                ops._TUNED[key + "#i3"] = min(i3, key=i3.get)
    print("%d of %d shapes changed; summed in-step gain %.3f ms per step (event-timed)" % (changed, len(times), gain))
    ops.save_tune_cache(args.out)


if __name__ == "__main__":
    main()
