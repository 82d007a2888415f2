"""Timeline of the overlapped weight gradients: three replayed steps under the ROCm tracer (torch.profiler); for the
last step prints per kernel class the summed time, and for each grouped weight-gradient launch its start / end relative
to the step start, beside the span of the backward chain.
usage: MBX_WG_OVERLAP=K [MBX_WG_GROUPS=n] python tools/overlap_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
from multibox_amd.trainer import Trainer, decay_steps
from multibox_amd import priors as PR
from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS
from torch.profiler import profile, ProfilerActivity
B = 64
priors = PR.priors_for_input_size(DEFAULT_ASPECT_RATIOS[5], 299).astype(np.float32)
net = Net(batch=B, input_size=299, k=5, mode="train", seed=2)
tr = Trainer(net, priors, max_num_bboxes=13, location_loss_alpha=1000.0, decay_steps_=decay_steps(56945, B, 4), use_graph=True)
images, gt, n = synthetic_batch(B, 299, 13, seed=0)
tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
ev = []
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        t0 = e.time_range.start
        ev.append((t0, t0 + float(getattr(e, "device_time", None) or e.cuda_time), e.name))
ev.sort()
starts = [i for i, e in enumerate(ev) if "pack_input_kernel" in e[2]]
a = starts[-1]
step = ev[a:]
T0 = step[0][0]
cls = {}
for s, e, nm in step:
    key = nm.split("(")[0].split("<")[0].replace("void ", "").replace("(anonymous namespace)::", "").strip()
    c = cls.setdefault(key, [0.0, 0])
    c[0] += e - s
    c[1] += 1
print("K=%d groups=%d: step wall %.3f ms, kernel time %.3f ms" % (net.wgrad_overlap_cus, len(tr._segments), (max(e for _, e, _ in step) - T0) / 1e3,
                                                                 sum(e - s for s, e, _ in step) / 1e3))
for k, (t, c) in sorted(cls.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %-48s x%-4d %9.1f us" % (k[:48], c, t))
bw = [(s, e) for s, e, nm in step if "head_scatter" in nm]
print("backward starts at %.3f ms" % ((bw[0][0] - T0) / 1e3 if bw else -1))
for s, e, nm in step:
    if "wgrad_grouped" in nm:
        print("  wgrad launch: %.3f -> %.3f ms (%.3f ms)" % ((s - T0) / 1e3, (e - T0) / 1e3, (e - s) / 1e3))
opt = [(s, e) for s, e, nm in step if "rmsprop" in nm]
if opt:
    print("optimiser starts at %.3f ms" % ((opt[0][0] - T0) / 1e3))
