"""Per-layer table of the detect forward (BASELINE config 4: 256 patches, k=7, folded BN): kernel time from the ROCm
tracer (torch.profiler) of graph-free replays, joined with the engine's launch order (inference: one conv launch per
ConvOp)."""
import os, sys, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net, ConvOp
from torch.profiler import profile, ProfilerActivity

B = int(os.environ.get("DB_B", "256"))
net = Net(batch=B, input_size=299, k=7, mode="infer")
net.fold_bn()
net.set_input(torch.rand(B, 299, 299, 3, device="cuda") * 2 - 1)
for _ in range(3):
    net.forward()
torch.cuda.synchronize()
REP = 5
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(REP):
        net.forward()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
ev.sort(key=lambda e: e.time_range.start)
CONV = ("conv_igemm", "conv_direct", "conv_resident", "conv_stem", "conv_pwres")
convs = [e for e in ev if any(k in e.name for k in CONV)]
ops = [op for op in net.fwd if isinstance(op, ConvOp)]
assert len(convs) == REP * len(ops), (len(convs), len(ops))
other = {}
for e in ev:
    if not any(k in e.name for k in CONV):
        k = re.sub(r"<.*|\(.*", "", e.name)
        other[k] = other.get(k, 0.0) + float(e.device_time) / REP
rows = {}
for i, e in enumerate(convs):
    op = ops[i % len(ops)]
    name = re.sub(r"_\d+/", "_N/", op.name.replace("InceptionResnetV2/", ""))
    key = (name[:58], op.M, op.Cin, op.K, op.R, op.S)
    r = rows.setdefault(key, [0.0, 0, ""])
    r[0] += float(e.device_time) / REP
    r[1] += 1
    m = re.search(r"conv_(\w+?)_kernel<([^>]*)>", e.name)
    r[2] = ("%s:" % m.group(1)) + m.group(2) if m else e.name[:30]
tot = tf = 0.0
print("%-58s %8s %5s %5s %3s | %8s %6s %5s | kernel" % ("layer", "M", "Cin", "K", "RS", "us", "TF/s", "x"))
for (name, M, Ci, K, R, S), (us, n, kern) in rows.items():
    n //= REP
    fl = 2.0 * M * Ci * K * R * S * n
    tot += us; tf += fl
    print("%-58s %8d %5d %5d %dx%d | %8.1f %6.0f %5d | %s" % (name, M, Ci, K, R, S, us, fl / us / 1e6, n, kern))
print("conv total %.1f us, %.0f TFLOP/s; other kernels:" % (tot, tf / tot / 1e6))
for k, v in sorted(other.items(), key=lambda kv: -kv[1]):
    print("   %-50s %8.1f us" % (k, v))
