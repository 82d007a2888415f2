"""Diagnostics behind tests/test_gpu_model.py::test_full_depth_backward_teacher_forced: engine vs the bf16-emulating
and the float32 torch oracle, full depth, batch 8 -- per-variable cosine / rel-L2 of the gradients and of the forward
outputs; free-running and TEACHER-FORCED (the oracle takes the engine's stored activations at every layer boundary).
usage: python tools/teacher_forced_explore.py [batch] [frozen|train]"""
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
    from tests.test_gpu_model import oracle_params, rel_l2, _cos
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    # frozen: the ORACLES normalise with fixed statistics (bn_training=False) -- shows that the decorrelation of the free-running
    # gradients is not a batch-statistics effect; the engine has no such mode, so only oracle-vs-oracle lines mean anything then
    frozen = (sys.argv[2] if len(sys.argv) > 2 else "train") == "frozen"
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
    with torch.no_grad():
        cal = Model(P0, k=5, bn_training=True, q=q_bf16, bn_decay=0.0)
        cal.build(images)
    for scope, (mean, var) in cal.new_moving.items():
        net.set_param(scope + "/BatchNorm/moving_mean", mean)
        net.set_param(scope + "/BatchNorm/moving_variance", var)
        P0[scope + "/BatchNorm/moving_mean"], P0[scope + "/BatchNorm/moving_variance"] = mean.clone(), var.clone()
    net.set_input(images.cuda())
    net.forward()
    ml = MultiboxLoss(priors, B, 13, 1000.0)
    ml.d_locs, ml.d_logits = net.d_locs, net.d_logits
    ml.forward_backward(net.locs, net.logits, torch.from_numpy(gt).cuda(), torch.from_numpy(n_gt).cuda())
    net.zero_grads()
    net.backward()
    torch.cuda.synchronize()
    match = ml.match.cpu().numpy()
    from tests.test_gpu_model import engine_activations
    force = engine_activations(net)
    res = {}
    for tag, q, fo in (("q", q_bf16, None), ("f32", None, None), ("tf", q_bf16, force)):
        P = {k_: v.clone().requires_grad_(True) for k_, v in P0.items()}
        m = Model(P, k=5, bn_training=not frozen, q=q, force=fo)
        rl, rz = m.build(images if q else images.to(torch.bfloat16).float())
        loc, conf = multibox_loss(rl, rz, torch.from_numpy(priors), torch.from_numpy(gt), match, 1000.0)
        (loc + conf).backward()
        res[tag] = (P, m, rl.detach(), rz.detach())
    print("forward rel-L2 locs: engine-q %.4f  q-f32 %.4f   logits: engine-q %.4f  q-f32 %.4f" % (
        rel_l2(net.locs.cpu(), res["q"][2]), rel_l2(res["q"][2], res["f32"][2]),
        rel_l2(net.logits.cpu(), res["q"][3]), rel_l2(res["q"][3], res["f32"][3])))
    for k, v in net.endpoints.items():
        print("  endpoint %-18s engine-q %.4f   q-f32 %.4f" % (k, rel_l2(v.tensor().float().cpu().permute(0, 3, 1, 2), res["q"][1].endpoints[k].detach()),
                                                            rel_l2(res["q"][1].endpoints[k].detach(), res["f32"][1].endpoints[k].detach())))
    names = [n for n in net.param_index if n.endswith(("/weights", "/biases", "/beta"))]
    gq = {n: res["q"][0][n].grad for n in names}
    gf = {n: res["f32"][0][n].grad for n in names}
    ge = {n: net.get_param(n, "grad").detach().float().cpu() for n in names}
    med = np.median([float(gq[n].norm()) for n in names])
    big = [n for n in names if float(gq[n].norm()) > 1e-3 * med]
    ce = np.array([_cos(ge[n], gq[n]) for n in big]); ci = np.array([_cos(gq[n], gf[n]) for n in big])
    le = np.array([rel_l2(ge[n], gq[n]) for n in big]); li = np.array([rel_l2(gq[n], gf[n]) for n in big])
    print("%d of %d variables; cosine engine-q: min %.4f p5 %.4f median %.4f | q-f32: min %.4f p5 %.4f median %.4f" % (
        len(big), len(names), ce.min(), np.percentile(ce, 5), np.median(ce), ci.min(), np.percentile(ci, 5), np.median(ci)))
    print("rel-L2 engine-q: max %.4f p95 %.4f median %.4f | q-f32: max %.4f p95 %.4f median %.4f" % (
        le.max(), np.percentile(le, 95), np.median(le), li.max(), np.percentile(li, 95), np.median(li)))
    order = np.argsort(ce)
    for i in order[:12]:
        print("  worst: %-70s cos %.4f (q-f32 %.4f) relL2 %.3f (q-f32 %.3f) |g| %.3g" % (big[i][-70:], ce[i], ci[i], le[i], li[i], float(gq[big[i]].norm())))
    gt_ = {n: res["tf"][0][n].grad for n in names}
    ct = np.array([_cos(ge[n], gt_[n]) for n in big]); lt = np.array([rel_l2(ge[n], gt_[n]) for n in big])
    print("TEACHER-FORCED oracle: cosine min %.5f p5 %.5f median %.5f | rel-L2 max %.4f p95 %.4f median %.4f" % (
        ct.min(), np.percentile(ct, 5), np.median(ct), lt.max(), np.percentile(lt, 95), np.median(lt)))
    for i in np.argsort(ct)[:10]:
        print("  tf worst: %-70s cos %.5f relL2 %.4f |g| %.3g" % (big[i][-70:], ct[i], lt[i], float(gt_[big[i]].norm())))
    whole_t = torch.cat([gt_[n].reshape(-1) for n in names])
    whole_e = torch.cat([ge[n].reshape(-1) for n in names]); whole_q = torch.cat([gq[n].reshape(-1) for n in names]); whole_f = torch.cat([gf[n].reshape(-1) for n in names])
    print("whole gradient vs teacher-forced oracle: cos %.6f rel-L2 %.4f" % (_cos(whole_e, whole_t), rel_l2(whole_e, whole_t)))
    print("whole gradient: cos engine-q %.5f  q-f32 %.5f   rel-L2 engine-q %.4f  q-f32 %.4f" % (
        _cos(whole_e, whole_q), _cos(whole_q, whole_f), rel_l2(whole_e, whole_q), rel_l2(whole_q, whole_f)))


if __name__ == "__main__":
    main()
