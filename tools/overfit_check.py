import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g; g.build()
from multibox_amd.engine import Net
from multibox_amd.trainer import Trainer
from multibox_amd import priors as PR
from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS
B = 8
pri = np.array(PR.generate_priors(DEFAULT_ASPECT_RATIOS[5]), np.float32)
net = Net(batch=B, input_size=299, k=5, mode="train", seed=3)
tr = Trainer(net, pri, max_num_bboxes=13, use_graph=True, initial_learning_rate=0.01)
images, gt, n = synthetic_batch(B, 299, 13, seed=5)
tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())
for it in range(400):
    tr.step()
    if it % 40 == 0 or it == 399:
        l = tr.losses()
        print(it, "loc %.2f conf %.2f reg %.3f total %.2f" % l, "status", int(tr.match_status().max()))
