import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
from multibox_amd.trainer import Trainer
from multibox_amd import priors as PR
B = 2
gen = torch.Generator().manual_seed(3)
images = torch.rand(B, 299, 299, 3, generator=gen) * 2 - 1
priors = np.array(PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]), np.float32)
rng = np.random.RandomState(1)
n_gt = np.array([3, 0], np.int32); gt = np.zeros((B, 13, 4), np.float32)
for b in range(B):
    xy = rng.uniform(0, .7, (n_gt[b], 2)); wh = rng.uniform(.05, .3, (n_gt[b], 2)); gt[b, :n_gt[b], :2] = xy; gt[b, :n_gt[b], 2:] = xy + wh
junk = [torch.full((64 * 1024 * 1024,), 1e30, device="cuda") for _ in range(8)]   # dirty the allocator's cache
del junk
outs = []
for trial, use_graph in enumerate([False, False, True]):
    net = Net(batch=B, input_size=299, k=5, mode="train", seed=5)
    tr = Trainer(net, priors, max_num_bboxes=13, use_graph=use_graph, n_segments=3)
    tr.set_batch(images.cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n_gt).cuda())
    tr.step(); torch.cuda.synchronize()
    ep = {k: v.tensor().float().cpu().clone() for k, v in net.endpoints.items()}
    print("trial", trial, "graph" if use_graph else "eager", "loss", tr.losses(), "finite", bool(torch.isfinite(net.locs).all()),
          "match", torch.nonzero(tr.loss.match[0] >= 0).flatten().tolist())
    outs.append((ep, net.locs.cpu().clone(), net.logits.cpu().clone(), net.Wg.cpu().clone(), net.W.cpu().clone()))
    tr.step(); torch.cuda.synchronize(); print("   step2 loss", tr.losses())
    del net, tr
for i in (1, 2):
    print("trial 0 vs", i, {k: bool(torch.equal(outs[0][0][k], outs[i][0][k])) for k in outs[0][0]},
          "locs", bool(torch.equal(outs[0][1], outs[i][1])), "logits", bool(torch.equal(outs[0][2], outs[i][2])),
          "Wg rel", float((outs[0][3] - outs[i][3]).norm() / outs[0][3].norm()), "W rel", float((outs[0][4] - outs[i][4]).norm() / outs[0][4].norm()))
