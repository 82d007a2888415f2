"""How far ahead of the GPU the host runs in a data-parallel step (one rank, MBX_FORCE_DIST=1, RCCL): host time to ENQUEUE a
step against the GPU time of the step.  usage: MBX_FORCE_DIST=1 python tools/dp_host_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29677")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch
import torch.distributed as dist
import __graft_entry__ as g
g.build()
torch.cuda.set_device(0)
pg = None
if os.environ.get("MBX_FORCE_DIST"):
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    pg = dist.group.WORLD
from multibox_amd.engine import Net
from multibox_amd.trainer import Trainer, decay_steps
from multibox_amd import priors as PR
from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS
B = 64
priors = PR.priors_for_input_size(DEFAULT_ASPECT_RATIOS[5], 299).astype(np.float32)
net = Net(batch=B, input_size=299, k=5, mode="train", seed=2, bn_max_workgroups=192 if pg is not None else 0)
tr = Trainer(net, priors, max_num_bboxes=13, location_loss_alpha=1000.0, decay_steps_=decay_steps(56945, B, 4), use_graph=True, process_group=pg)
images, gt, n = synthetic_batch(B, 299, 13, seed=0)
tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter()
    tr.step()
    host.append((time.perf_counter() - a) * 1e3)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("segments %d, reducer %s: host enqueue %.2f ms per step (median %.2f, max %.2f), GPU step %.2f ms" % (
    len(tr._segments), tr.reducer.enabled, t_enq / 20 * 1e3, sorted(host)[10], max(host), t_all / 20 * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    tr.step(); torch.cuda.synchronize()
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
ev = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
starts = [i for i, e in enumerate(ev) if "pack_input" in e.name]
seg = ev[starts[1]:starts[2]]                      # one whole step
busy = sum(float(e.device_time) for e in seg)
span = seg[-1].time_range.end - seg[0].time_range.start
gaps = []
for a, b in zip(seg, seg[1:]):
    gp = b.time_range.start - a.time_range.end
    if gp > 3.0:
        gaps.append((round(gp, 1), a.name[:40], b.name[:40]))
gaps.sort(reverse=True)
print("one step: %d kernels, busy %.2f ms, span %.2f ms, idle %.2f ms; gaps > 3 us: %d, the largest: %s" % (len(seg), busy / 1e3, span / 1e3, (span - busy) / 1e3, len(gaps), gaps[:8]))
cls = {}
for e in seg:
    k = next((c for c in ("conv_igemm", "conv_wgrad", "bn_bwd", "bn_apply", "bn_finalize", "pool", "rmsprop", "Memcpy", "fill", "Fill") if c in e.name), "other")
    o = cls.setdefault(k, [0, 0.0]); o[0] += 1; o[1] += float(e.device_time)
print({k: (v[0], round(v[1] / 1e3, 3)) for k, v in sorted(cls.items())})
names = {}
for e in seg:
    if not any(k in e.name for k in ("conv_", "bn_", "pool", "rmsprop", "match", "loss", "head_", "pack_", "filter_prepare", "ema_")):
        names[e.name[:60]] = names.get(e.name[:60], 0.0) + float(e.device_time)
print({k: round(v, 1) for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:8]})
if pg is not None:
    dist.destroy_process_group()
