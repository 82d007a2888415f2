"""End-to-end GPU parity of the engine (forward, loss, backward, optimizer step) against the
torch-CPU oracle model run on the SAME bf16-rounded weights, with its activations rounded where the
engine stores bf16 (oracle/torch_model.py `q`).

Stated tolerances (bf16 storage, fp32 accumulate, ~100 layers deep):
  forward: backbone features relative L2 error < 2e-2; locations/logits max abs error < 5e-2 * max|ref|;
  matching: indices identical to the oracle's on the ENGINE's own outputs (bit-exact integer work);
  losses: rtol 1e-5 vs the numpy oracle on the engine's outputs;
  weight gradients: relative L2 error per tensor < 6e-2 (gradients cross bf16 at every layer).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available()
    from multibox_amd.engine import Net
    from multibox_amd import priors as PR
    torch.manual_seed(0)
    B = 2
    net = Net(batch=B, input_size=299, k=5, mode="train")
    gen = torch.Generator().manual_seed(3)
    # non-trivial BN betas so that relu masks are exercised
    net.Bt.copy_((torch.randn(net.nBt, generator=gen) * 0.1).cuda())
    images = torch.rand(B, 299, 299, 3, generator=gen) * 2 - 1
    priors = np.array(PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]), np.float32)
    rng = np.random.RandomState(1)
    n_gt = np.array([3, 0], np.int32)
    gt = np.zeros((B, 13, 4), np.float32)
    for b in range(B):
        xy = rng.uniform(0, .7, (n_gt[b], 2)); wh = rng.uniform(.05, .3, (n_gt[b], 2))
        gt[b, :n_gt[b], :2] = xy; gt[b, :n_gt[b], 2:] = xy + wh
    return dict(torch=torch, net=net, images=images, priors=priors, gt=gt, n_gt=n_gt, B=B)


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def oracle_params(torch, net):
    P = {}
    for name in net.param_index:
        v = net.get_param(name).detach().float().cpu().clone()
        if name.endswith("/weights"):
            v = v.to(torch.bfloat16).float()          # the engine's kernels read the bf16 copy
        P[name] = v
    return P


def test_forward_backward_parity(env):
    torch, net = env["torch"], env["net"]
    from oracle.torch_model import Model, q_bf16, multibox_loss
    from oracle import ref_numpy as R
    from multibox_amd.loss import MultiboxLoss
    P = oracle_params(torch, net)
    for v in P.values():
        v.requires_grad_(True)
    mm0 = net.MM.clone()
    net.set_input(env["images"].cuda())
    locs, logits = net.forward()
    torch.cuda.synchronize()
    locs, logits = locs.cpu(), logits.cpu()
    m = Model(P, k=5, bn_training=True, q=q_bf16)
    rl, rz = m.build(env["images"])
    # ---- forward
    f_eng = net.features.tensor().float().cpu().permute(0, 3, 1, 2)
    e = rel_l2(f_eng, m.endpoints["Conv2d_7b_1x1"].detach())
    assert e < 2e-2, "features rel L2 %.4f" % e
    assert float((locs - rl.detach()).abs().max()) < 5e-2 * float(rl.abs().max()), "locations"
    assert float((logits - rz.detach()).abs().max()) < 5e-2 * float(rz.abs().max()), "logits"
    # moving statistics were updated like slim.batch_norm does
    name = "InceptionResnetV2/Conv2d_2b_3x3"
    mm_new, mv_new = m.new_moving[name]
    assert torch.allclose(net.get_param(name + "/BatchNorm/moving_mean").cpu(), mm_new, rtol=2e-2, atol=1e-5)
    assert torch.allclose(net.get_param(name + "/BatchNorm/moving_variance").cpu(), mv_new, rtol=2e-2, atol=1e-5)
    assert not torch.equal(mm0, net.MM)
    # ---- loss on the engine's outputs vs the numpy oracle (integer work exact, sums rtol 1e-5)
    ml = MultiboxLoss(env["priors"], env["B"], 13, 1000.0)
    ml.d_locs, ml.d_logits = net.d_locs, net.d_logits
    loss2, _, _ = ml.forward_backward(net.locs, net.logits, torch.from_numpy(env["gt"]).cuda(), torch.from_numpy(env["n_gt"]).cuda())
    torch.cuda.synchronize()
    ref = R.add_loss(locs.numpy(), R.sigmoid_f32(logits.numpy()), env["gt"], env["n_gt"], env["priors"], 1000.0)
    assert np.array_equal(ml.match.cpu().numpy(), ref["match"])
    l2 = loss2.cpu().numpy()
    assert np.isclose(l2[0], ref["loc_loss"], rtol=1e-5) and np.isclose(l2[1], ref["conf_loss"], rtol=1e-5)
    # ---- backward: same matching fed to the torch oracle
    net.zero_grads()
    net.backward()
    torch.cuda.synchronize()
    loc, conf = multibox_loss(rl, rz, torch.from_numpy(env["priors"]), torch.from_numpy(env["gt"]), ref["match"], 1000.0)
    (loc + conf).backward()
    worst = []
    checked = 0
    for name in net.param_index:
        if not (name.endswith("/weights") or name.endswith("/biases") or name.endswith("/beta")):
            continue
        g_ref = P[name].grad
        g_eng = net.get_param(name, "grad").detach().float().cpu()
        if float(g_ref.norm()) < 1e-12:
            continue
        worst.append((rel_l2(g_eng, g_ref), name))
        checked += 1
    worst.sort(reverse=True)
    assert checked > 400
    assert worst[0][0] < 6e-2, "worst gradient mismatches: %s" % worst[:8]


def test_train_steps_graph_equals_eager(env):
    torch = env["torch"]
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    res = []
    for use_graph in (False, True):
        net = Net(batch=2, input_size=299, k=5, mode="train", seed=5)
        tr = Trainer(net, env["priors"], max_num_bboxes=13, use_graph=use_graph, n_segments=3)
        tr.set_batch(env["images"].cuda(), torch.from_numpy(env["gt"]).cuda(), torch.from_numpy(env["n_gt"]).cuda())
        w0 = net.W.clone()
        losses = []
        for _ in range(3):
            tr.step()
            losses.append(tr.losses())
        torch.cuda.synchronize()
        assert int(tr.match_status().max()) == 0
        assert all(np.isfinite(x) for l_ in losses for x in l_)
        assert not torch.equal(w0, net.W)
        assert losses[0][2] > 0 and abs(losses[0][3] - sum(losses[0][:3])) < 1e-3 * abs(losses[0][3])
        res.append((losses, net.W.clone(), tr.Wema.clone()))
    # graph replay runs the same kernels on the same data; only fp32 atomics ordering may differ
    for a, b in zip(res[0][0], res[1][0]):
        assert np.allclose(a, b, rtol=2e-2), (a, b)
    assert rel_l2(res[1][1], res[0][1]) < 1e-3


def test_finetune_and_infer_modes(env):
    torch = env["torch"]
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    from oracle.torch_model import Model, q_bf16
    net = Net(batch=2, input_size=299, k=5, mode="train", fine_tune=True, seed=6)
    tr = Trainer(net, env["priors"], max_num_bboxes=13, use_graph=False)
    tr.set_batch(env["images"].cuda(), torch.from_numpy(env["gt"]).cuda(), torch.from_numpy(env["n_gt"]).cuda())
    w0 = net.W.clone()
    tr.step()
    torch.cuda.synchronize()
    assert torch.equal(w0[:net.head_w_start], net.W[:net.head_w_start])          # backbone frozen (train.py:229-232)
    assert not torch.equal(w0[net.head_w_start:], net.W[net.head_w_start:])
    # inference mode (detect.py:313-334): all BN frozen, parity with the oracle in eval mode
    inf = Net(batch=2, input_size=299, k=5, mode="infer", seed=6)
    inf.fold_bn()
    inf.set_input(env["images"].cuda())
    locs, logits = inf.forward()
    P = oracle_params(torch, inf)
    with torch.no_grad():
        rl, rz = Model(P, k=5, bn_training=False, q=q_bf16).build(env["images"])
    assert float((locs.cpu() - rl).abs().max()) < 5e-2 * float(rl.abs().max())
    assert float((logits.cpu() - rz).abs().max()) < 5e-2 * float(rz.abs().max())
