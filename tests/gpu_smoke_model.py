"""smoke() part 2: one tiny training step of the full network on cuda:0 (forward, on-device matching +
loss checked against the numpy oracle on the engine's own outputs, backward, RMSProp/EMA)."""
import numpy as np


def run():
    import torch
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    from multibox_amd import priors as PR
    from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS
    from oracle import ref_numpy as R
    B = 2
    priors = np.array(PR.generate_priors(DEFAULT_ASPECT_RATIOS[5]), np.float32)
    net = Net(batch=B, input_size=299, k=5, mode="train")
    tr = Trainer(net, priors, max_num_bboxes=13, use_graph=False)
    images, gt, n = synthetic_batch(B, 299, 13, seed=0)
    n[0] = max(int(n[0]), 2)
    gt[0, :2] = [[0.1, 0.1, 0.5, 0.6], [0.3, 0.4, 0.9, 0.8]]
    tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())
    w0 = net.W.clone()
    tr.step()
    torch.cuda.synchronize()
    loc, conf, reg, total = tr.losses()
    ref = R.add_loss(net.locs.cpu().numpy(), R.sigmoid_f32(net.logits.cpu().numpy()), gt, n, priors, 1000.0)
    assert int(tr.match_status().max()) == 0
    assert np.array_equal(tr.loss.match.cpu().numpy(), ref["match"]), "match indices differ from the oracle"
    assert np.isclose(loc, ref["loc_loss"], rtol=1e-5) and np.isclose(conf, ref["conf_loss"], rtol=1e-5)
    assert np.isfinite(total) and not torch.equal(w0, net.W)
    print("model smoke ok: total_loss=%.3f (loc %.3f conf %.3f reg %.4f), P=%d" % (total, loc, conf, reg, net.P))
