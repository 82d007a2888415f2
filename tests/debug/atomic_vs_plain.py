"""Debug: the same B=2 network with atomic statistics rows and with plain rows + finalize, one process: per-layer
batch statistics, activations and gradients side by side (scratch script, not collected by pytest)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
from multibox_amd import priors as PR
from multibox_amd.loss import MultiboxLoss

B = 2
gen = torch.Generator().manual_seed(3)
betas = None
images = None
priors = np.array(PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]), np.float32)
rng = np.random.RandomState(1)
n_gt = np.array([3, 0], np.int32)
gt = np.zeros((B, 13, 4), np.float32)
for b in range(B):
    xy = rng.uniform(0, .7, (n_gt[b], 2)); wh = rng.uniform(.05, .3, (n_gt[b], 2))
    gt[b, :n_gt[b], :2] = xy; gt[b, :n_gt[b], 2:] = xy + wh
res = {}
for mode in ("1", "0"):
    os.environ["MBX_ATOMIC_STATS"] = mode
    torch.manual_seed(0)
    net = Net(batch=B, input_size=299, k=5, mode="train")
    if betas is None:
        betas = torch.randn(net.nBt, generator=gen) * 0.1
        images = torch.rand(B, 299, 299, 3, generator=gen) * 2 - 1
    net.Bt.copy_(betas.cuda())
    net.set_input(images.cuda())
    net.forward()
    ml = MultiboxLoss(priors, B, 13, 1000.0)
    ml.d_locs, ml.d_logits = net.d_locs, net.d_logits
    ml.forward_backward(net.locs, net.logits, torch.from_numpy(gt).cuda(), torch.from_numpy(n_gt).cuda())
    net.zero_grads()
    net.backward()
    torch.cuda.synchronize()
    r = {"mean": net.bn_mean.cpu().clone(), "rstd": net.bn_rstd.cpu().clone(), "Wg": net.Wg.cpu().clone(), "Btg": net.Btg.cpu().clone(),
         "locs": net.locs.cpu().clone(), "logits": net.logits.cpu().clone(), "tiles": [(repr(k)[:60], d.tile_config) for k, d, w in net.tune_registry]}
    acts = {}
    for op in net.convs:
        if op.kind == "bn":
            acts[op.name] = (op.y_view.tensor().float().cpu().clone(), op.out.tensor().float().cpu().clone(), op.beta_off, op.K)
    r["acts"] = acts
    res[mode] = r
a, p = res["1"], res["0"]
def rl2(x, y):
    return float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30))
print("locs", rl2(a["locs"], p["locs"]), "logits", rl2(a["logits"], p["logits"]), "Wg", rl2(a["Wg"], p["Wg"]), "Btg", rl2(a["Btg"], p["Btg"]))
nt = sum(1 for x, y in zip(a["tiles"], p["tiles"]) if x != y)
print("tile choices that differ:", nt, [(x, y) for x, y in zip(a["tiles"], p["tiles"]) if x != y][:6])
for name, (ya, oa, bo, K) in a["acts"].items():
    yp, op_, _, _ = p["acts"][name]
    dm = rl2(a["mean"][bo:bo + K], p["mean"][bo:bo + K]); dr = rl2(a["rstd"][bo:bo + K], p["rstd"][bo:bo + K])
    dy, do = rl2(ya, yp), rl2(oa, op_)
    dg = rl2(a["Btg"][bo:bo + K], p["Btg"][bo:bo + K])
    if max(dm, dr, dy, do) > 1e-3 or dg > 2e-2:
        print("%-60s y %.2e a %.2e mean %.2e rstd %.2e dbeta %.2e" % (name[-60:], dy, do, dm, dr, dg))
print("done")
