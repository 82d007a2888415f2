"""GPU debugging aid: determinism of forward, effect of backward on a following forward,
and per-endpoint error of the engine against the torch-CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
from multibox_amd.trainer import Trainer
from multibox_amd import priors as PR
from oracle.torch_model import Model, q_bf16

def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))

B = 2
net = Net(batch=B, input_size=299, k=5, mode="train", seed=5)
gen = torch.Generator().manual_seed(3)
images = torch.rand(B, 299, 299, 3, generator=gen) * 2 - 1
priors = np.array(PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]), np.float32)
rng = np.random.RandomState(1)
n_gt = np.array([3, 0], np.int32); gt = np.zeros((B, 13, 4), np.float32)
for b in range(B):
    xy = rng.uniform(0, .7, (n_gt[b], 2)); wh = rng.uniform(.05, .3, (n_gt[b], 2)); gt[b, :n_gt[b], :2] = xy; gt[b, :n_gt[b], 2:] = xy + wh
tr = Trainer(net, priors, max_num_bboxes=13, use_graph=False)
tr.set_batch(images.cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n_gt).cuda())
W0, Wb0, Bt0 = net.W.clone(), net.Wb.clone(), net.Bt.clone()
def snap():
    torch.cuda.synchronize()
    return {k: v.tensor().float().cpu().clone() for k, v in net.endpoints.items()}, net.locs.cpu().clone(), net.logits.cpu().clone()
tr._front(); e1, l1, z1 = snap(); print("loss A", tr.loss.loss2.tolist(), "match", torch.nonzero(tr.loss.match[0] >= 0).flatten().tolist())
tr._front(); e2, l2, z2 = snap(); print("loss A'", tr.loss.loss2.tolist())
print("fwd deterministic:", all(torch.equal(e1[k], e2[k]) for k in e1), torch.equal(l1, l2), torch.equal(z1, z2))
for fns, lo, hi in tr._segments:
    for f in fns: f()
torch.cuda.synchronize()
print("weights untouched by backward:", torch.equal(W0, net.W), torch.equal(Wb0, net.Wb), torch.equal(Bt0, net.Bt))
tr._front(); e3, l3, z3 = snap(); print("loss after bwd", tr.loss.loss2.tolist(), "match", torch.nonzero(tr.loss.match[0] >= 0).flatten().tolist())
for k in e1:
    print("  fwd-after-bwd equal %-16s %s  rel %.3g" % (k, torch.equal(e1[k], e3[k]), rel(e3[k], e1[k])))
print("locs equal", torch.equal(l1, l3), "logits equal", torch.equal(z1, z3))
# per-endpoint error vs oracle
P = {}
for name in net.param_index:
    v = net.get_param(name).detach().float().cpu().clone()
    if name.endswith("/weights"): v = v.to(torch.bfloat16).float()
    P[name] = v
with torch.no_grad():
    m = Model(P, k=5, bn_training=True, q=q_bf16); rl, rz = m.build(images)
for k in e1:
    print("  vs oracle %-16s rel L2 %.4f" % (k, rel(e1[k].permute(0, 3, 1, 2), m.endpoints[k])))
print("locs max err %.4g / %.4g ; logits %.4g / %.4g" % (float((l1 - rl).abs().max()), float(rl.abs().max()), float((z1 - rz).abs().max()), float(rz.abs().max())))
