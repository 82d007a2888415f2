"""Exploration: full-depth gradient agreement engine vs torch oracle as a function of the batch size (a larger batch
makes the random-init BN network far less chaotic than the batch-2 test).  Prints per-stage cosine / rel-L2 statistics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
from multibox_amd.loss import MultiboxLoss
from multibox_amd import priors as PR
from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS
from oracle.torch_model import Model, q_bf16, multibox_loss
from oracle import ref_numpy as R

B = int(os.environ.get("GP_B", "16"))
torch.set_num_threads(int(os.environ.get("GP_THREADS", "16")))
net = Net(batch=B, input_size=299, k=5, mode="train", seed=2)
gen = torch.Generator().manual_seed(3)
net.Bt.copy_((torch.randn(net.nBt, generator=gen) * 0.1).cuda())
pri = np.array(PR.generate_priors(DEFAULT_ASPECT_RATIOS[5]), np.float32)
images, gt, n_gt = synthetic_batch(B, 299, 13, seed=0)
P0 = {}
for name in net.param_index:
    v = net.get_param(name).detach().float().cpu().clone()
    P0[name] = v.to(torch.bfloat16).float() if name.endswith("/weights") else v
net.set_input(torch.from_numpy(images).cuda())
locs, logits = net.forward()
ml = MultiboxLoss(pri, B, 13, 1000.0)
ml.d_locs, ml.d_logits = net.d_locs, net.d_logits
ml.forward_backward(net.locs, net.logits, torch.from_numpy(gt).cuda(), torch.from_numpy(n_gt).cuda())
net.zero_grads(); net.backward(); torch.cuda.synchronize()
ref = R.add_loss(locs.cpu().numpy(), R.sigmoid_f32(logits.cpu().numpy()), gt, n_gt, pri, 1000.0)
names = [n for n in net.param_index if n.endswith(("/weights", "/biases", "/beta"))]
ge = {n: net.get_param(n, "grad").detach().float().cpu() for n in names}
res = {}
for tag, q in (("q", q_bf16), ("f32", None)):
    t0 = time.time()
    P = {k_: v.clone().requires_grad_(True) for k_, v in P0.items()}
    m = Model(P, k=5, bn_training=True, q=q)
    x = torch.from_numpy(images)
    rl, rz = m.build(x if q else x.to(torch.bfloat16).float())
    loc, conf = multibox_loss(rl, rz, torch.from_numpy(pri), torch.from_numpy(gt), ref["match"], 1000.0)
    (loc + conf).backward()
    res[tag] = {n: P[n].grad for n in names}
    print("oracle", tag, "%.0f s" % (time.time() - t0), flush=True)
cos = lambda a, b: float((a.double().flatten() @ b.double().flatten()) / (a.double().norm() * b.double().norm() + 1e-30))
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
def stage(n):
    for key in ("Multibox", "Block8", "Repeat_2", "Mixed_7a", "Repeat_1", "Mixed_6a", "Repeat/", "Mixed_5b", "Conv2d_7b"):
        if key in n: return key
    return "stem"
rows = {}
for n in names:
    if float(res["q"][n].norm()) == 0: continue
    rows.setdefault(stage(n), []).append((cos(ge[n], res["q"][n]), rel(ge[n], res["q"][n]), cos(res["q"][n], res["f32"][n]), rel(res["q"][n], res["f32"][n])))
print("B=%d  stage: n | engine-vs-oracle_bf16 cos median/min, relL2 median | oracle_bf16-vs-f32 cos median/min, relL2 median" % B)
for k, v in rows.items():
    a = np.array(v)
    print("%-10s %3d | %.4f %.4f %.3f | %.4f %.4f %.3f" % (k, len(v), np.median(a[:, 0]), a[:, 0].min(), np.median(a[:, 1]), np.median(a[:, 2]), a[:, 2].min(), np.median(a[:, 3])))
allq = torch.cat([res["q"][n].flatten() for n in names]); alle = torch.cat([ge[n].flatten() for n in names]); allf = torch.cat([res["f32"][n].flatten() for n in names])
print("whole gradient: engine-vs-bf16 cos %.5f relL2 %.4f | bf16-vs-f32 cos %.5f relL2 %.4f" % (cos(alle, allq), rel(alle, allq), cos(allq, allf), rel(allq, allf)))
