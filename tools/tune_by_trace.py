"""Tile choice of every forward / data-gradient convolution launch from the REPLAYED step's own kernel timestamps.

tools/tune_in_situ.py times launches of an EAGER step with HIP events: good enough to rank tiles of MFMA-bound shapes,
but the event markers and the eager launch gaps change what sits in L2 / Infinity Cache between two launches, and on the
HBM-bound shapes (residual "up" convolutions, fused 1x1 data gradients) the ranking can come out wrong -- the detect
configuration's first igemm7 picks measured 675 -> 238 us by events and 637 -> 662 us in the replayed graph.  This tool
measures what bench.py measures: for every candidate configuration the step's hipGraphs are re-captured with that
configuration on every launch it applies to, replayed, and each launch's duration is read from the ROCm tracer
(torch.profiler: the timestamps rocprofv3 reports).  A challenger replaces the current choice of a layer shape only if
it is faster by --threshold; the new table is then verified on the step's wall time before it is written.

usage: python tools/tune_by_trace.py [--batch 64] [--input-size 299] [--k 5] [--max-num-bboxes 13] [--infer | --fine-tune] [--out FILE]
"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CANDIDATES = (2, 4, 5, 6, 9, 10, 12, 13, 14, 33, 34, 35, 36, 37, 38, 39, 65, 99)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--input-size", type=int, default=299)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--max-num-bboxes", type=int, default=13)
    ap.add_argument("--infer", action="store_true", help="the detect path: inference-mode forward only (BATCH_SIZE patches)")
    ap.add_argument("--fine-tune", action="store_true", help="train.py --fine_tune: frozen backbone, the heads train")
    ap.add_argument("--steps", type=int, default=4, help="traced replays per candidate")
    ap.add_argument("--threshold", type=float, default=0.015)
    ap.add_argument("--out", default=None)
    ap.add_argument("--dry", action="store_true", help="measure and report, do not write the table")
    ap.add_argument("--candidates", default=None, help="comma-separated tile configurations to try instead of the full list")
    args = ap.parse_args()
    global CANDIDATES
    if args.candidates:
        CANDIDATES = tuple(int(v) for v in args.candidates.split(","))
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    from torch.profiler import profile, ProfilerActivity
    from multibox_amd import _lib, ops, priors as PR
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS

    l = _lib.lib()
    if args.infer:
        net = Net(batch=args.batch, input_size=args.input_size, k=args.k, mode="infer")
        net.fold_bn()
        imgs = torch.rand(args.batch, args.input_size, args.input_size, 3, device="cuda") * 2 - 1
        graph = {"g": None}

        def recapture():
            net.set_input(imgs)
            net.forward()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                net.set_input(imgs)                            # (pack_input: the marker of a step's start in the trace)
                net.forward()
            graph["g"] = gr

        def step():
            graph["g"].replay()
    else:
        net = Net(batch=args.batch, input_size=args.input_size, k=args.k, mode="train", fine_tune=args.fine_tune, seed=2)
        pri = PR.priors_for_input_size(DEFAULT_ASPECT_RATIOS[args.k], args.input_size).astype(np.float32)
        tr = Trainer(net, pri, max_num_bboxes=args.max_num_bboxes, location_loss_alpha=1000.0, use_graph=True)
        images, gt, n = synthetic_batch(args.batch, args.input_size, args.max_num_bboxes, seed=0)
        tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())

        def recapture():
            tr.graphs = None
            tr.step()

        step = tr.step

    reg = [(key, d, what) for key, d, what in net.tune_registry if what in ("fwd", "dgrad")]
    by_addr = {C.addressof(d): (i, key, d) for i, (key, d, _) in enumerate(reg)}
    tuned = [d.tile_config for _, d, _ in reg]
    counters0 = [d.work_counter for _, d, _ in reg]
    # every launch gets its own block of work counters for the candidate passes (igemm7 wants 32)
    big = torch.zeros(max(len(reg) * ops.I7_COUNTERS, 8), dtype=torch.int32, device="cuda")
    net.i5_counters, net._i5_used = big, big.numel()
    # What Net.forward() clears at the start of every pass must cover BOTH the tool's counters and the statistics rows the
    # convolutions ADD into (net.stats16, a view of the net's own _fwd_clear): with only `big` swapped in, the rows were never
    # zeroed and every candidate pass normalised by statistics accumulated over all the passes before it.
    class _ClearBoth:
        def __init__(self, a, b): self.a, self.b = a, b
        def zero_(self): self.a.zero_(); self.b.zero_(); return self
    net._fwd_clear = _ClearBoth(big, net._fwd_clear)
    orig_conv, orig_bn = l.mbx_conv, l.mbx_bn_apply_fused
    state = {"cand": None, "rows": None, "calls": []}

    def conv(desc_ref, stream):
        d = desc_ref._obj
        ent = by_addr.get(C.addressof(d))
        if ent is None:                                        # (the float32 head launches: not tuned, but they are in the trace)
            state["calls"].append((-1, d.tile_config))
            return orig_conv(desc_ref, stream)
        i = ent[0]
        want = state["cand"][i] if isinstance(state["cand"], list) else (state["cand"] or tuned[i])
        d.tile_config = want
        d.work_counter = big.data_ptr() + 4 * ops.I7_COUNTERS * i if want > ops.I5_FLAG else None
        r = orig_conv(desc_ref, stream)
        if r != 0:                                             # the candidate does not apply to this launch
            d.tile_config = want = tuned[i]
            d.work_counter = big.data_ptr() + 4 * ops.I7_COUNTERS * i if want > ops.I5_FLAG else None
            r = orig_conv(desc_ref, stream)
        if d.stats_partial:
            state["rows"] = ops.conv_stats_rows(d)            # the partial-row count follows the tile
        state["calls"].append((i, want))
        return r

    def bn_apply_fused(stats, rows, *rest):
        return orig_bn(stats, state["rows"] if state["rows"] is not None else rows, *rest)

    def measure(cand):
        """Re-capture with `cand` (None: the current table; int: that configuration everywhere it applies; list: per
        launch) and return ({launch index: (cfg, mean us)}, wall ms per step)."""
        state.update(cand=cand, rows=None, calls=[])
        recapture()
        torch.cuda.synchronize()
        calls = state["calls"]
        n_calls = len(calls) // 2                              # eager warm-up pass + captured pass
        calls = calls[-n_calls:]
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()                                             # (the tracer may miss the first kernels after it starts)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / args.steps * 1e3
        ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        ev.sort(key=lambda e: e.time_range.start)
        steps, cur = [], None                                  # the conv launches of each step: a step starts at pack_input
        for e in ev:
            if "pack_input" in e.name:
                cur = []
                steps.append(cur)
            elif any(k_ in e.name for k_ in ("conv_igemm", "conv_direct", "conv_stem", "conv_resident", "conv_pwres")) and "pair_kernel" not in e.name \
                    and cur is not None:                       # (pair launches go through mbx_conv_pair: not tuned here)
                cur.append(e)
        steps = [s_ for s_ in steps if len(s_) == n_calls]
        assert len(steps) >= max(args.steps - 1, 1), ([len(s_) for s_ in steps], n_calls)
        out = {}
        for s_ in steps:
            for j, e in enumerate(s_):
                i, cfg = calls[j]
                o = out.setdefault((i, j), [cfg, 0.0])
                o[1] += float(e.device_time) / len(steps)
        return out, wall

    l.mbx_conv, l.mbx_bn_apply_fused = conv, bn_apply_fused
    try:
        base, wall0 = measure(None)
        times = {}                                             # key -> cfg -> summed us of all its launches
        for (i, _), (cfg, us) in base.items():
            if i < 0:
                continue
            times.setdefault(reg[i][0], {}).setdefault(cfg, 0.0)
            times[reg[i][0]][cfg] += us
        for c in CANDIDATES:
            res, wall = measure(c)
            acc = {}
            for (i, _), (cfg, us) in res.items():
                if i >= 0 and cfg == c:
                    acc[reg[i][0]] = acc.get(reg[i][0], 0.0) + us
            for key, us in acc.items():
                if c not in times[key]:
                    times[key][c] = us
                else:
                    times[key][c] = min(times[key][c], us)
            print("candidate %2d: applies to %3d shapes, step %.3f ms" % (c, len(acc), wall), flush=True)
        cur_of = {}
        for (key, d, _), t in zip(reg, tuned):
            cur_of[key] = t
        new = dict(cur_of)
        changed, gain = 0, 0.0
        for key, per in times.items():
            cur = cur_of[key]
            if cur not in per:
                continue
            best = min(per, key=per.get)
            if best != cur and per[best] < per[cur] * (1.0 - args.threshold):
                new[key] = best
                changed += 1
                gain += per[cur] - per[best]
                print("%-118s %2d -> %2d  %8.1f -> %8.1f us" % (key[:118], cur, best, per[cur], per[best]))
        print("%d of %d shapes would change; summed gain %.3f ms per step by the kernels' timestamps" % (changed, len(times), gain * 1e-3))
        # verification on the wall clock: the new table against the old one, alternating
        per_launch = [new[key] for key, _, _ in reg]
        walls = {"old": [], "new": []}
        for _ in range(2):
            walls["old"].append(measure(None)[1])
            walls["new"].append(measure(per_launch)[1])
        print("step wall time: old table %s ms, new table %s ms" % (["%.3f" % w for w in walls["old"]], ["%.3f" % w for w in walls["new"]]))
        better = min(walls["new"]) < min(walls["old"]) and sum(walls["new"]) < sum(walls["old"])
        if changed and better and not args.dry:
            rule = (ops.DIRECT3_TILE_CONFIG, ops.DIRECTW_TILE_CONFIG, ops.RESIDENT_TILE_CONFIG, ops.PWRES_TILE_CONFIG)
            for key, cfg in new.items():
                if cfg > ops.SPLITK_FLAG or cfg in rule:
                    continue                                  # chosen by rule at net build (split-K, direct launches): not table entries
                if cur_of[key] in rule:
                    ops._TUNED[key + "#norule"] = cfg         # a table tile beat the rule's launch: the measured exception (Net._tune)
                ops._TUNED[key] = cfg
                if cfg > ops.I5_FLAG:
                    i3 = {c: t for c, t in times[key].items() if 0 < c <= ops.N_TILE_CONFIGS}
                    if i3:
                        ops._TUNED[key + "#i3"] = min(i3, key=i3.get)
            ops.save_tune_cache(args.out)
            print("table written")
        else:
            print("table NOT written (changed %d, better %s, dry %s)" % (changed, better, args.dry))
    finally:
        l.mbx_conv, l.mbx_bn_apply_fused = orig_conv, orig_bn
        for (_, d, _), t, w in zip(reg, tuned, counters0):
            d.tile_config, d.work_counter = t, w


if __name__ == "__main__":
    main()
