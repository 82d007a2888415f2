"""Overlapped weight gradients (Trainer, net.wgrad_overlap_cus) against the plain step: MBX_DETERMINISTIC=1, the same
batch and seed, N steps each -> parameters, EMA shadows and gradients must be bit-identical (same kernels, same work
items; only where and when the grouped weight-gradient launches run differs).
usage: MBX_DETERMINISTIC=1 python tools/overlap_check.py [K] [steps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MBX_DETERMINISTIC", "1")
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
from multibox_amd.trainer import Trainer, decay_steps
from multibox_amd import priors as PR
from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS
K = int(sys.argv[1]) if len(sys.argv) > 1 else 96
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
priors = PR.priors_for_input_size(DEFAULT_ASPECT_RATIOS[5], 299).astype(np.float32)
images, gt, n = synthetic_batch(B, 299, 13, seed=0)


def run(k):
    net = Net(batch=B, input_size=299, k=5, mode="train", seed=2, wgrad_overlap_cus=k)
    tr = Trainer(net, priors, max_num_bboxes=13, location_loss_alpha=1000.0, decay_steps_=decay_steps(56945, B, 4), use_graph=True)
    tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    assert tr.check_health() == {"stop": False, "fallback": False}
    out = [t.clone() for t in (net.W, net.Bt, net.MM, net.MV, tr.Wema, net.Wg, net.Btg)]
    print("K=%d: %d segments, graphs %d, losses %s" % (k, len(tr._segments), len(tr.graphs), tr.losses()), flush=True)
    del tr, net
    torch.cuda.empty_cache()
    return out


a, b = run(0), run(K)
names = ("W", "Bt", "MM", "MV", "Wema", "Wg", "Btg")
bad = [nm for nm, x, y in zip(names, a, b) if not torch.equal(x, y)]
for nm, x, y in zip(names, a, b):
    print("%-5s max|diff| %.3e" % (nm, float((x - y).abs().max())))
print("OVERLAP_CHECK", "FAIL " + ",".join(bad) if bad else "OK bit-identical")
sys.exit(1 if bad else 0)
