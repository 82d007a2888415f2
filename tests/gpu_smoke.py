"""smoke(): one small invocation of the hot path on cuda:0, checked against the oracle."""
import numpy as np


def run():
    import torch
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    from multibox_amd import loss as L, priors as PR
    from oracle import ref_numpy as R
    priors = np.array(PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]), np.float32)
    rng = np.random.RandomState(0)
    B, P, G = 4, priors.shape[0], 13
    raw = (rng.randn(B, P, 4) * 0.05).astype(np.float32)
    logits = (rng.randn(B, P) * 2 - 2).astype(np.float32)
    n = np.array([5, 0, 13, 1], np.int32)
    gt = np.zeros((B, G, 4), np.float32)
    for b in range(B):
        xy = rng.uniform(0, .7, (n[b], 2)); wh = rng.uniform(.05, .3, (n[b], 2))
        gt[b, :n[b], :2] = xy; gt[b, :n[b], 2:] = xy + wh
    ml = L.MultiboxLoss(priors, B, G, 1000.0)
    loss2, dl, dz = ml.forward_backward(torch.from_numpy(raw).cuda(), torch.from_numpy(logits).cuda(),
                                        torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())
    torch.cuda.synchronize()
    ref = R.add_loss(raw, R.sigmoid_f32(logits), gt, n, priors, 1000.0)
    assert np.array_equal(ml.match.cpu().numpy(), ref["match"]), "match indices differ from the oracle"
    l2 = loss2.cpu().numpy()
    assert np.isclose(l2[0], ref["loc_loss"], rtol=1e-5) and np.isclose(l2[1], ref["conf_loss"], rtol=1e-5)
    print("smoke ok: loc_loss=%.4f conf_loss=%.4f (oracle %.4f %.4f)" % (l2[0], l2[1], ref["loc_loss"], ref["conf_loss"]))
    try:
        from tests import gpu_smoke_model
    except ImportError:
        return
    gpu_smoke_model.run()


if __name__ == "__main__":
    run()
