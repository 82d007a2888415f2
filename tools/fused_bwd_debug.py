"""Debug: per-layer dy (and trunk gradients) of the fused backward pass against the unfused one, same network / batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
from multibox_amd import priors as PR
from multibox_amd.loss import MultiboxLoss
B = int(os.environ.get("KB_B", "8"))
gen = torch.Generator().manual_seed(31)
images = (torch.rand(B, 299, 299, 3, generator=gen) * 2 - 1).cuda()
beta = (torch.randn(200000, generator=gen) * 0.1).cuda()
priors = np.array(PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]), np.float32)
rng = np.random.RandomState(5)
n_gt = np.array(([3, 0, 13, 1, 5, 2, 7, 4] * 8)[:B], np.int32)
gt = np.zeros((B, 13, 4), np.float32)
for b in range(B):
    xy = rng.uniform(0, .7, (n_gt[b], 2)); wh = rng.uniform(.05, .3, (n_gt[b], 2))
    gt[b, :n_gt[b], :2] = xy; gt[b, :n_gt[b], 2:] = xy + wh


def run(fuse):
    os.environ["MBX_RESIDENT_MIN_IMAGES"] = "1"
    os.environ["MBX_FUSE_BWD"] = fuse
    net = Net(batch=B, input_size=299, k=5, mode="train", seed=5)
    net.Bt.copy_(beta[:net.nBt])
    net.zero_grads(); net.set_input(images); net.forward()
    ml = MultiboxLoss(priors, B, 13, 1000.0)
    ml.d_locs, ml.d_logits = net.d_locs, net.d_logits
    ml.forward_backward(net.locs, net.logits, torch.from_numpy(gt).cuda(), torch.from_numpy(n_gt).cuda())
    net.backward(); torch.cuda.synchronize()
    out = {}
    for op in net.convs:
        if op.kind == "bn" and op.trainable:
            out[op.name] = (op.dy_view.tensor().float().cpu().clone(), op.fb_unit is not None,
                            getattr(op, "fdesc", None).tile_config if getattr(op, "fdesc", None) is not None else None)
    return out, net


a, na = run("1")
b, nb = run("0")
c, nc = run("0")
rel = lambda x, y: float((x - y).norm() / (y.norm() + 1e-30))
print("%-70s %9s %9s %s" % ("layer (backward order)", "fused/unf", "unf/unf", "fused? colsum-ratio"))
for name in reversed(list(a)):
    x, y, z = a[name][0], b[name][0], c[name][0]
    e1, e2 = rel(x, y), rel(z, y)
    flag = " <<<" if e1 > max(5e-3, 4 * e2) else ""
    cs = float(x.reshape(-1, x.shape[-1]).sum(0).abs().max()), float(y.reshape(-1, y.shape[-1]).sum(0).abs().max())
    print("%-70s %9.2e %9.2e %s colsum %.3g vs %.3g%s" % (name[-70:], e1, e2, a[name][1], cs[0], cs[1], flag))
