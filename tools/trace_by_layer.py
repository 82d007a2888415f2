"""Join a rocprofv3 kernel trace of bench.py with the engine's launch order: per-layer time of the
conv forward / data-gradient / weight-gradient kernels in the LAST step of the trace.
usage: python tools/trace_by_layer.py gpurun_out/prof/x_kernel_trace.csv [batch]"""
import csv, sys, os, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multibox_amd.engine import Net, ConvOp

path = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rows = [r for r in csv.DictReader(open(path))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last step: from the last pack_input kernel to the end
starts = [i for i, n in enumerate(names) if "pack_input" in n]
step = rows[starts[-1]:]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ig = [r for r in step if any(k in r["Kernel_Name"] for k in ("conv_igemm3", "conv_igemm5", "conv_igemm7", "conv_direct3", "conv_directw", "conv_resident", "conv_pwres", "conv_stem"))]      # igemm5: the persistent launches
wg = [r for r in step if "conv_wgrad2" in r["Kernel_Name"]]
net = Net(batch=B, device="cpu")
convs = [op for op in net.fwd if isinstance(op, ConvOp)]
bw = [op for op in reversed(net.fwd) if isinstance(op, ConvOp) and op.trainable]
# Walk the igemm launches in order.  A batch-norm group of two may run as ONE pair launch (conv_igemm3_pair_kernel, issued at
# the position of the group's last member, forward, and of its first member in backward order): its time is split between
# the two convolutions by their FLOPs.
kname = lambda r: ("pair:" if "pair_kernel" in r["Kernel_Name"] else "i5:" if "igemm5" in r["Kernel_Name"] else "i7:" if "igemm7" in r["Kernel_Name"] else "d3:" if "direct3" in r["Kernel_Name"] else "dw:" if "directw" in r["Kernel_Name"] else "res:" if "conv_resident" in r["Kernel_Name"] else "pw:" if "conv_pwres" in r["Kernel_Name"] else "stem:" if "conv_stem" in r["Kernel_Name"] else "") + re.search(r"<([^>]*)>", r["Kernel_Name"]).group(1)
flops = lambda op: 2.0 * op.M * op.K * op.R * op.S * op.Cin
rec = {}
pos = 0
for op in convs:
    g = getattr(op, "group", None)
    r = ig[pos]
    if g is not None and len(g.members) == 2 and "pair_kernel" in r["Kernel_Name"]:
        if op is g.members[0]:
            continue                                   # its launch is the pair at the last member's position
        a, b = g.members
        fa, fb = flops(a), flops(b)
        for m, f in ((a, fa), (b, fb)):
            rec[id(m)] = {"op": m, "fwd": dur(r) * f / (fa + fb), "fwd_k": kname(r), "fwd_grid": r["Grid_Size_X"]}
        pos += 1
        continue
    rec[id(op)] = {"op": op, "fwd": dur(r), "fwd_k": kname(r), "fwd_grid": r["Grid_Size_X"]}
    pos += 1
dg = ig[pos:]
dgi = iter(dg)
if len(wg) == len(bw):          # per-layer weight-gradient launches (round 1); the grouped launches have no per-layer time
    for op, r in zip(bw, wg):
        rec[id(op)]["wg"] = dur(r)
        rec[id(op)]["wg_grid"] = "%sx%s" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"])
paired = set()
for op in bw:
    if op.need_dx:
        if id(op) in paired:
            continue
        d = next(dgi)
        g = getattr(op, "group", None)
        if g is not None and len(g.members) == 2 and "pair_kernel" in d["Kernel_Name"]:
            a, b = g.members
            fa, fb = flops(a), flops(b)
            for m, f in ((a, fa), (b, fb)):
                rec[id(m)]["dg"] = dur(d) * f / (fa + fb)
                rec[id(m)]["dg_k"] = kname(d)
                paired.add(id(m))
            continue
        rec[id(op)]["dg"] = dur(d)
        rec[id(op)]["dg_k"] = kname(d)
print("%-58s %7s %5s %5s %3s | %7s %7s %7s | TF/s fwd dg wg | %s" % ("layer", "M", "Cin", "K", "RS", "fwd us", "dg us", "wg us", "cfg"))
agg = {}
tot = [0, 0, 0]
for op in convs:
    r = rec[id(op)]
    fl = 2.0 * op.M * op.K * op.R * op.S * op.Cin
    f, d, w = r["fwd"], r.get("dg", 0), r.get("wg", 0)
    tot[0] += f; tot[1] += d; tot[2] += w
    tf = lambda t: fl / t / 1e6 if t else 0
    name = re.sub(r"_\d+/", "_N/", op.name.replace("InceptionResnetV2/", ""))[:58]
    key = (name, op.M, op.Cin, op.K, op.R, op.S)
    a = agg.setdefault(key, [0, 0, 0, 0, fl, r["fwd_k"], r.get("dg_k", ""), r.get("wg_grid", "")])
    a[0] += f; a[1] += d; a[2] += w; a[3] += 1
for key, a in agg.items():
    name, M, Cin, K, R, S = key
    n = a[3]
    tf = lambda t: a[4] * n / t / 1e6 if t else 0
    print("%-58s %7d %5d %5d %dx%d | %7.1f %7.1f %7.1f | %4.0f %4.0f %4.0f | x%d f<%s> d<%s> w[%s]" % (
        name, M, Cin, K, R, S, a[0], a[1], a[2], tf(a[0]), tf(a[1]), tf(a[2]), n, a[5], a[6], a[7]))
print("totals us: fwd %.0f dgrad %.0f wgrad %.0f" % tuple(tot))
other = {}
for r in step:
    n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("<")[0].split("(")[0]
    o = other.setdefault(n, [0, 0.0]); o[0] += 1; o[1] += dur(r)
print("kernels of the last step:")
for n, (c, t) in sorted(other.items(), key=lambda kv: -kv[1][1]):
    print("  %-50s x%-4d %9.1f us" % (n[:50], c, t))
print("  sum %.1f us; span %.1f us" % (sum(t for _, t in other.values()), (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3))
