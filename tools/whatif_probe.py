"""What-if timing probes: upper bounds of what a restructuring could save, measured on the real step by REMOVING the
work in question (results are wrong during the probe; shapes, launch order and data statistics stay).
  apply   the bn_apply launches of the block35 / block17 / block8 layers skipped (their `a` buffers keep the previous
          step's values) = the most a consumer-side batch-norm apply could save
  epi     residual forward epilogues and accumulate / mask data-gradient epilogues replaced by a plain store = the most
          an epilogue fully hidden behind the next tile's K loop could save
  bnbwd   the batch-norm backward launches skipped
usage: python tools/whatif_probe.py apply|epi|bnbwd"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
from multibox_amd.trainer import Trainer, decay_steps
from multibox_amd import priors as PR, ops
from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS
mode = sys.argv[1] if len(sys.argv) > 1 else "apply"
B = 64
priors = PR.priors_for_input_size(DEFAULT_ASPECT_RATIOS[5], 299).astype(np.float32)
net = Net(batch=B, input_size=299, k=5, mode="train", seed=2)
tr = Trainer(net, priors, max_num_bboxes=13, location_loss_alpha=1000.0, decay_steps_=decay_steps(56945, B, 4), use_graph=True)
images, gt, n = synthetic_batch(B, 299, 13, seed=0)
tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())


def timed(k=30):
    for _ in range(5):
        tr.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        tr.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3


a = timed()
saved = []
if mode == "apply":
    net._probe_skip_apply = True
elif mode == "bnbwd":
    net._probe_skip_bn_bwd = True
elif mode == "epi":
    for _, d, what in net.tune_registry:
        if d.epilogue == ops.EPI_RESIDUAL or (what == "dgrad" and (d.accumulate or d.skip)):
            saved.append((d, d.epilogue, d.accumulate, d.skip, d.acc_src, d.relu))
            d.epilogue, d.accumulate, d.skip, d.acc_src, d.relu = ops.EPI_STORE, 0, None, None, 0
tr.graphs = None
b = timed()
net._probe_skip_apply = net._probe_skip_bn_bwd = False
for d, e, ac, sk, asrc, rl in saved:
    d.epilogue, d.accumulate, d.skip, d.acc_src, d.relu = e, ac, sk, asrc, rl
tr.graphs = None
c = timed()
print("%s: normal %.3f ms, probe %.3f ms (%d descriptors changed), normal again %.3f ms" % (mode, a, b, len(saved), c))
