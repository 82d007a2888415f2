"""Where does the teacher-forced gradient comparison lose agreement in the stem?  Engine da (gradient wrt a layer's output)
and d(beta) against the oracle's, layer by layer.  usage: python tools/stem_grad_probe.py [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MBX_DETERMINISTIC"] = "1"


def main():
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    from multibox_amd.engine import Net
    from multibox_amd import priors as PR
    from multibox_amd.loss import MultiboxLoss
    from oracle.torch_model import Model, q_bf16, multibox_loss
    from tests.test_gpu_model import oracle_params, rel_l2, _cos, engine_activations
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    net = Net(batch=B, input_size=299, k=5, mode="train", seed=5)
    gen = torch.Generator().manual_seed(11)
    net.Bt.copy_((torch.randn(net.nBt, generator=gen) * 0.1).cuda())
    images = torch.rand(B, 299, 299, 3, generator=gen) * 2 - 1
    priors = np.array(PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]), np.float32)
    rng = np.random.RandomState(4)
    n_gt = np.array(([3, 0, 13, 1, 5, 2, 7, 4] * B)[:B], np.int32)
    gt = np.zeros((B, 13, 4), np.float32)
    for b in range(B):
        xy = rng.uniform(0, .7, (n_gt[b], 2)); wh = rng.uniform(.05, .3, (n_gt[b], 2))
        gt[b, :n_gt[b], :2] = xy; gt[b, :n_gt[b], 2:] = xy + wh
    P0 = oracle_params(torch, net)
    net.set_input(images.cuda())
    net.forward()
    ml = MultiboxLoss(priors, B, 13, 1000.0)
    ml.d_locs, ml.d_logits = net.d_locs, net.d_logits
    ml.forward_backward(net.locs, net.logits, torch.from_numpy(gt).cuda(), torch.from_numpy(n_gt).cuda())
    net.zero_grads()
    net.backward()
    torch.cuda.synchronize()
    match = ml.match.cpu().numpy()
    P = {k_: v.clone().requires_grad_(True) for k_, v in P0.items()}
    m = Model(P, k=5, bn_training=True, q=q_bf16, force=engine_activations(net))
    m.keep_acts = True
    rl, rz = m.build(images)
    loc, conf = multibox_loss(rl, rz, torch.from_numpy(priors), torch.from_numpy(gt), match, 1000.0)
    (loc + conf).backward()
    for op in net.convs[:12]:
        if op.kind != "bn":
            continue
        off = 0
        for mem in op.members:
            sc = mem.scope
            a = m.acts[sc]
            da_o = a.grad                                                             # [B,C,H,W]
            da_e = net.grad_of(op.out).slice(off, mem.K).tensor().float().cpu().permute(0, 3, 1, 2)
            db_o = P[sc + "/BatchNorm/beta"].grad
            db_e = net.get_param(sc + "/BatchNorm/beta", "grad").float().cpu()
            mask = (a.detach() > 0).float()
            print("%-60s da: cos %.5f relL2 %.4f | dbeta: cos %.5f relL2 %.4f | sum(da*mask) oracle-check %.4f engine-da %.4f | nnz(da) %.3f" % (
                sc[-60:], _cos(da_e, da_o), rel_l2(da_e, da_o), _cos(db_e, db_o), rel_l2(db_e, db_o),
                rel_l2((da_o * mask).sum((0, 2, 3)), db_o), rel_l2((da_e * mask).sum((0, 2, 3)), db_o), float((da_o != 0).float().mean())))
            off += mem.K


if __name__ == "__main__":
    main()
