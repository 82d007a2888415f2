import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
from multibox_amd.loss import MultiboxLoss
from multibox_amd import priors as PR
from oracle.torch_model import Model, q_bf16, multibox_loss
from oracle import ref_numpy as R
def rel(a, b):
    a, b = a.double(), b.double(); return float((a - b).norm() / (b.norm() + 1e-30))
B = int(os.environ.get("DBG_B", "2"))
net = Net(batch=B, input_size=299, k=5, mode="train", seed=5)
gen = torch.Generator().manual_seed(3)
net.Bt.copy_((torch.randn(net.nBt, generator=gen) * 0.1).cuda())
images = torch.rand(B, 299, 299, 3, generator=gen) * 2 - 1
priors = np.array(PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]), np.float32)
rng = np.random.RandomState(1)
n_gt = rng.randint(0, 6, B).astype(np.int32); n_gt[0] = 3
gt = np.zeros((B, 13, 4), np.float32)
for b in range(B):
    xy = rng.uniform(0, .7, (n_gt[b], 2)); wh = rng.uniform(.05, .3, (n_gt[b], 2)); gt[b, :n_gt[b], :2] = xy; gt[b, :n_gt[b], 2:] = xy + wh
def params(q):
    P = {}
    for name in net.param_index:
        v = net.get_param(name).detach().float().cpu().clone()
        if name.endswith("/weights") and q: v = v.to(torch.bfloat16).float()
        P[name] = v.requires_grad_(True)
    return P
net.set_input(images.cuda()); net.forward()
ml = MultiboxLoss(priors, B, 13, 1000.0); ml.d_locs, ml.d_logits = net.d_locs, net.d_logits
ml.forward_backward(net.locs, net.logits, torch.from_numpy(gt).cuda(), torch.from_numpy(n_gt).cuda())
net.zero_grads(); net.backward(); torch.cuda.synchronize()
match = ml.match.cpu().numpy()
res = {}
for tag, q in (("q", q_bf16), ("f32", None)):
    P = params(True)
    m = Model(P, k=5, bn_training=True, q=q)
    rl, rz = m.build(images if q else images.to(torch.bfloat16).float())
    loc, conf = multibox_loss(rl, rz, torch.from_numpy(priors), torch.from_numpy(gt), match, 1000.0)
    (loc + conf).backward()
    res[tag] = (P, m, rl.detach(), rz.detach())
print("B", B, "n_gt", n_gt.tolist())
for k in net.endpoints:
    e = net.endpoints[k].tensor().float().cpu().permute(0, 3, 1, 2)
    print("  %-16s eng-vs-q %.4f   q-vs-f32 %.4f" % (k, rel(e, res["q"][1].endpoints[k].detach()), rel(res["q"][1].endpoints[k].detach(), res["f32"][1].endpoints[k].detach())))
print("locs: eng-vs-q max %.4f, q-vs-f32 max %.4f (max|ref| %.3f)" % (float((net.locs.cpu() - res["q"][2]).abs().max()), float((res["q"][2] - res["f32"][2]).abs().max()), float(res["q"][2].abs().max())))
rows = []
for name in net.param_index:
    if not name.endswith(("/weights", "/biases", "/beta")): continue
    gq, gf = res["q"][0][name].grad, res["f32"][0][name].grad
    ge = net.get_param(name, "grad").detach().float().cpu()
    if float(gq.norm()) < 1e-12: continue
    rows.append((rel(ge, gq), rel(gq, gf), name, float(gq.norm())))
print("Wg abs max", float(net.Wg.abs().max()), "Btg abs max", float(net.Btg.abs().max()), "d_locs max", float(net.d_locs.abs().max()))
for nm in ["Multibox/8x8/Conv_2/weights", "Multibox/8x8/Conv_1/weights", "Multibox/1x1/Conv/weights", "InceptionResnetV2/Conv2d_7b_1x1/weights", "InceptionResnetV2/Block8/Conv2d_1x1/weights", "InceptionResnetV2/Conv2d_2a_3x3/weights", "Multibox/8x8/Conv_1/BatchNorm/beta"]:
    ge = net.get_param(nm, "grad").detach().float().cpu(); gq = res["q"][0][nm].grad; gf = res["f32"][0][nm].grad
    print("  %-50s |eng| %.4g |q| %.4g |f32| %.4g  cos(eng,q) %.4f cos(q,f32) %.4f" % (nm, float(ge.norm()), float(gq.norm()), float(gf.norm()),
          float((ge*gq).sum()/(ge.norm()*gq.norm()+1e-30)), float((gf*gq).sum()/(gf.norm()*gq.norm()+1e-30))))
rows.sort(reverse=True)
print("worst engine-vs-q gradient errors (eng-vs-q, q-vs-f32, name, |g|):")
for r in rows[:12]: print("   %.4f %.4f %s %.3g" % r)
a = np.array([r[0] for r in rows]); b_ = np.array([r[1] for r in rows])
print("n=%d eng-vs-q: median %.4f p90 %.4f max %.4f | q-vs-f32: median %.4f p90 %.4f max %.4f" % (len(rows), np.median(a), np.percentile(a, 90), a.max(), np.median(b_), np.percentile(b_, 90), b_.max()))
c1 = []; c2 = []
for r in rows:
    ge = net.get_param(r[2], "grad").detach().float().cpu(); gq = res["q"][0][r[2]].grad; gf = res["f32"][0][r[2]].grad
    c1.append(float((ge*gq).sum()/(ge.norm()*gq.norm()+1e-30))); c2.append(float((gf*gq).sum()/(gf.norm()*gq.norm()+1e-30)))
c1 = np.array(c1); c2 = np.array(c2)
print("cosine eng~q: min %.3f p10 %.3f median %.3f | q~f32: min %.3f p10 %.3f median %.3f | min(c1-c2) %.3f" % (c1.min(), np.percentile(c1,10), np.median(c1), c2.min(), np.percentile(c2,10), np.median(c2), (c1-c2).min()))
print("ratio eng/q-noise: median %.3f p90 %.3f max %.3f" % (np.median(a / b_), np.percentile(a / b_, 90), (a / b_).max()))
tot_e = torch.cat([net.get_param(r[2], "grad").detach().float().cpu().reshape(-1) for r in rows]); tot_q = torch.cat([res["q"][0][r[2]].grad.reshape(-1) for r in rows]); tot_f = torch.cat([res["f32"][0][r[2]].grad.reshape(-1) for r in rows])
print("whole-gradient rel L2: eng-vs-q %.4f, q-vs-f32 %.4f" % (rel(tot_e, tot_q), rel(tot_q, tot_f)))
