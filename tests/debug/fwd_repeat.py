"""Debug: forward pass repeated on one net: is it bit-reproducible at batch 2 (where every statistics row has at most two adders)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
B = int(os.environ.get("KB_B", "2"))
torch.manual_seed(0)
net = Net(batch=B, input_size=299, k=5, mode="train")
gen = torch.Generator().manual_seed(3)
net.Bt.copy_((torch.randn(net.nBt, generator=gen) * 0.1).cuda())
images = torch.rand(B, 299, 299, 3, generator=gen) * 2 - 1
net.set_input(images.cuda())
outs = []
for i in range(6):
    net.forward()
    torch.cuda.synchronize()
    outs.append((net.locs.clone(), net.logits.clone(), net.bn_mean.clone(), net.bn_rstd.clone()))
for i in range(1, 6):
    d = [float((a.double() - b.double()).abs().max()) for a, b in zip(outs[0], outs[i])]
    # first layer whose mean differs
    bad = None
    for op in net.convs:
        if op.kind == "bn":
            sl = slice(op.beta_off, op.beta_off + op.K)
            if not torch.equal(outs[0][2][sl], outs[i][2][sl]):
                bad = op.name
                break
    print("run %d vs 0: max |d locs| %.3e |d logits| %.3e |d mean| %.3e |d rstd| %.3e first differing layer: %s" % (i, *d, bad))
