"""Diagnostic: train on one synthetic batch for many steps and report where the first non-finite value appears
(flat parameter buffer -> owning variable or alignment gap; gradient buffer; optimiser slots)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
from multibox_amd.trainer import Trainer, decay_steps
from multibox_amd import priors as PR
from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS

steps = int(os.environ.get("NH_STEPS", "1500"))
every = int(os.environ.get("NH_EVERY", "25"))
B = 64
pri = PR.priors_for_input_size(DEFAULT_ASPECT_RATIOS[5], 299).astype(np.float32)
net = Net(batch=B, input_size=299, k=5, mode="train", seed=2)
tr = Trainer(net, pri, max_num_bboxes=13, decay_steps_=decay_steps(56945, B, 4), use_graph=True)
images, gt, n = synthetic_batch(B, 299, 13, seed=0)
tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())


def owner(off):
    for name, (buf, o, shape, cpad) in net.param_index.items():
        if buf != "W":
            continue
        cnt = int(np.prod(shape[:-1] + (cpad,) if cpad is not None else shape))
        if o <= off < o + cnt:
            return "%s[+%d of %d]" % (name, off - o, cnt)
    return "GAP"


for it in range(steps):
    tr.step()
    if (it + 1) % every == 0:
        torch.cuda.synchronize()
        bad = {}
        for nm, t in (("W", net.W), ("Wg", net.Wg), ("Wms", tr.Wms), ("Wema", tr.Wema), ("Bt", net.Bt), ("Btg", net.Btg), ("MM", net.MM), ("MV", net.MV)):
            m = ~torch.isfinite(t)
            if bool(m.any()):
                bad[nm] = m.nonzero().flatten()[:6].tolist()
        l = tr.losses()
        print(it + 1, "losses %.1f %.1f reg %.4f" % (l[0], l[1], l[2]), "max|W| %.3g max|Wg| %.3g" % (float(net.W.abs().max()), float(net.Wg.abs().max())), bad if bad else "", flush=True)
        if bad:
            for nm, idx in bad.items():
                if nm in ("W", "Wg", "Wms", "Wema"):
                    print("  ", nm, [(i, owner(i)) for i in idx])
            break
print("timeouts", net.barrier_timeouts(), "status", int(tr.match_status().max()))
