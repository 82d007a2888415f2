"""Idle gaps between consecutive kernels of a rocprofv3 kernel trace, per training step (a step starts at
pack_input_kernel): wall time, summed kernel time, summed gaps, the largest gaps with the kernels around them.
torch's own kernels (the gradient clears: Wg.zero_() is a 240 MB fill) are left out of the step's kernel list, so they
show up as gaps -- pass --all to keep them.
usage: python tools/trace_gaps.py <kernel_trace.csv> [min_gap_us] [--all]"""
import csv
import sys

args = [a for a in sys.argv[1:] if a != "--all"]
keep_all = "--all" in sys.argv
rows = list(csv.DictReader(open(args[0])))
thr = float(args[1]) if len(args) > 1 else 30.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
starts = [i for i, e in enumerate(ev) if "pack_input_kernel" in e[2]]
print("%d kernels, %d steps" % (len(ev), len(starts)))
for si, a in enumerate(starts):
    b = starts[si + 1] if si + 1 < len(starts) else len(ev)
    step = [e for e in ev[a:b] if keep_all or ("at::native" not in e[2] and "rocclr" not in e[2])]
    if len(step) < 100:
        continue
    wall = (step[-1][1] - step[0][0]) / 1e3
    busy = sum(e[1] - e[0] for e in step) / 1e3
    gaps = [((s1 - e0) / 1e3, n0, n1) for (s0, e0, n0), (s1, e1, n1) in zip(step, step[1:])]
    pos = [g for g in gaps if g[0] > 0]
    big = sorted([g for g in gaps if g[0] > thr], reverse=True)
    print("step %d: %d kernels, wall %.2f ms, kernel time %.2f ms, gaps %.2f ms (median gap %.2f us); gaps > %.0f us: %s" % (
        si, len(step), wall / 1e3, busy / 1e3, sum(g[0] for g in pos) / 1e3, sorted(g[0] for g in pos)[len(pos) // 2], thr,
        [(round(g[0], 1), g[1][22:60], g[2][22:60]) for g in big[:4]]))
