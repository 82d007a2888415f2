"""Idle gaps between consecutive kernels of a rocprofv3 kernel trace: the largest ones with the kernels around them.
usage: python tools/trace_gaps.py <kernel_trace.csv> [min_gap_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    g = (s1 - e0) / 1e3
    if g > thr:
        gaps.append((g, n0[:60], n1[:60], s1))
print("%d kernels, %d gaps > %.0f us" % (len(ev), len(gaps), thr))
for g, a, b, s in gaps[-40:]:
    print("%8.1f us  after %-60s before %-60s" % (g, a, b))
tot = sum((s1 - e0) for (s0, e0, _), (s1, e1, _) in zip(ev, ev[1:]) if 0 < (s1 - e0) < 1e6) / 1e6
print("sum of all gaps < 1 ms: %.2f ms over the trace" % tot)
