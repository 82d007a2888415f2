"""GPU parity: libmbx matching / loss / decode kernels vs the oracle and the golden fixtures.

Bar: match indices, partitions, stacked gt, detection scores/boxes bit-exact; loss
values rtol 1e-5 (the reference's TF reduction order is un-vendored); gradients rtol 1e-4.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MATCH_CASES = ["b2_p646_g5", "b4_p646_g13", "b3_p904_g100", "b2_p646_g13_wide", "b8_p646_g13_rand", "sat_alpha1"]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _priors_for(golden, P):
    return golden.priors["k5_restrict" if P == 646 else "k7_restrict"].astype(np.float32)


@pytest.mark.parametrize("case", MATCH_CASES)
def test_match_golden_exact(torch_cuda, golden, case):
    torch = torch_cuda
    from multibox_amd import loss as L
    g = golden.matching
    raw, confs, gt, n = g[case + "_raw"], g[case + "_confs"], g[case + "_gt"], g[case + "_n"]
    B, P = raw.shape[:2]
    priors = _priors_for(golden, P)
    alpha = 1.0 if case == "sat_alpha1" else 1000.0
    dec = (raw + priors[None]).astype(np.float32)
    c = (confs + np.float32(1e-10)).astype(np.float32)
    part, stacked = L.compute_assignments(torch.from_numpy(dec).cuda().reshape(-1, 4), torch.from_numpy(c).cuda().reshape(-1),
                                          torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda(), B, alpha)
    assert part.dtype == torch.int32
    assert np.array_equal(part.cpu().numpy(), g[case + "_part"])
    assert stacked.cpu().numpy().tobytes() == g[case + "_stacked"].tobytes()


@pytest.mark.parametrize("B,P,G,seed", [(64, 646, 13, 0), (16, 904, 100, 1), (4, 3199, 100, 2), (3, 70, 5, 3), (2, 13, 13, 4)])
def test_match_vs_oracle_random(torch_cuda, B, P, G, seed):
    torch = torch_cuda
    from multibox_amd import loss as L
    from oracle import ref_numpy as R
    rng = np.random.RandomState(seed)
    dec = rng.uniform(0, 1, (B, P, 4)).astype(np.float32)
    c = (R.sigmoid_f32(rng.randn(B, P) * 2 - 1) + np.float32(1e-10)).astype(np.float32)
    n = rng.randint(0, min(G, P) + 1, B).astype(np.int32)
    n[0] = min(G, P)
    if B > 1:
        n[1] = 0
    gt = np.zeros((B, G, 4), np.float32)
    for b in range(B):
        gt[b, :n[b]] = rng.uniform(0, 1, (n[b], 4))
    _, _, m_ref = R.compute_assignments(dec.reshape(-1, 4), c.reshape(-1), gt, n, B, 1000.0)
    m, st = L.match_boxes(torch.from_numpy(dec).cuda(), torch.from_numpy(c).cuda(), torch.from_numpy(gt).cuda(),
                          torch.from_numpy(n).cuda(), 1000.0)
    assert st.cpu().numpy().max() == 0
    m = m.cpu().numpy()
    assert np.array_equal(m, m_ref)
    # size-independent properties: every gt matched once, distinct predictions
    for b in range(B):
        assert sorted(m[b][m[b] >= 0].tolist()) == list(range(n[b]))


def test_match_error_status(torch_cuda):
    torch = torch_cuda
    from multibox_amd import loss as L
    B, P, G = 3, 8, 12
    dec = torch.rand(B, P, 4, device="cuda")
    c = torch.rand(B, P, device="cuda") * 0.9 + 0.05
    gt = torch.rand(B, G, 4, device="cuda")
    n = torch.tensor([12, 2, 2], dtype=torch.int32, device="cuda")     # image 0: n > P
    dec[2, 3, 1] = float("nan")                                        # image 2: non-finite
    m, st = L.match_boxes(dec, c, gt, n, 1000.0)
    assert st.cpu().tolist() == [1, 0, 2]
    with pytest.raises(ValueError):
        L.compute_assignments(dec.reshape(-1, 4), c.reshape(-1), gt, n, B, 1000.0)


@pytest.mark.parametrize("case", ["b4_p646_g13", "b3_p904_g100", "sat_alpha1"])
def test_loss_fwd_bwd_vs_oracle(torch_cuda, golden, case):
    torch = torch_cuda
    from multibox_amd import loss as L
    from oracle import ref_numpy as R
    g = golden.matching
    raw, logits, gt, n = g[case + "_raw"], g[case + "_logits"], g[case + "_gt"], g[case + "_n"]
    B, P = raw.shape[:2]
    priors = _priors_for(golden, P)
    alpha = 1.0 if case == "sat_alpha1" else 1000.0
    ml = L.MultiboxLoss(priors, B, gt.shape[1], alpha)
    loss2, dl, dz = ml.forward_backward(torch.from_numpy(raw).cuda(), torch.from_numpy(logits).cuda(),
                                        torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())
    torch.cuda.synchronize()
    ref = R.add_loss(raw, R.sigmoid_f32(logits), gt, n, priors, alpha)
    assert np.array_equal(ml.match.cpu().numpy(), ref["match"])
    l2 = loss2.cpu().numpy()
    assert np.isclose(l2[0], ref["loc_loss"], rtol=1e-5) and np.isclose(l2[1], ref["conf_loss"], rtol=1e-5)
    rdl, rdz = R.add_loss_grads(raw, logits, gt, priors, alpha, ref["match"])
    assert np.allclose(dl.cpu().numpy(), rdl, rtol=1e-4, atol=1e-6)
    assert np.allclose(dz.cpu().numpy(), rdz, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,k,G,seed", [(64, 5, 13, 0), (64, 7, 100, 1)])
def test_loss_full_batch_vs_oracle_and_additivity(torch_cuda, golden, B, k, G, seed):
    """BASELINE sizes (64 images per GPU; P = 646 / 904): decode + match + loss + gradients against the numpy oracle,
    and the size-independent property the reference's batch-sum loss has (loss.py:100-101): the loss of the batch is
    the sum of the losses of its images, the gradients of an image do not depend on the others."""
    torch = torch_cuda
    from multibox_amd import loss as L
    from oracle import ref_numpy as R
    P = 129 * k + 1
    priors = _priors_for(golden, P)
    rng = np.random.RandomState(seed)
    raw = (rng.randn(B, P, 4) * 0.05).astype(np.float32)
    logits = (rng.randn(B, P) * 2 - 2).astype(np.float32)
    n = rng.randint(0, G + 1, B).astype(np.int32)
    n[0], n[1] = G, 0
    gt = np.zeros((B, G, 4), np.float32)
    for b in range(B):
        xy = rng.uniform(0, .7, (n[b], 2)); wh = rng.uniform(.05, .3, (n[b], 2))
        gt[b, :n[b], :2] = xy; gt[b, :n[b], 2:] = xy + wh
    ml = L.MultiboxLoss(priors, B, G, 1000.0)
    loss2, dl, dz = ml.forward_backward(torch.from_numpy(raw).cuda(), torch.from_numpy(logits).cuda(),
                                        torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())
    torch.cuda.synchronize()
    assert int(ml.status.max()) == 0
    ref = R.add_loss(raw, R.sigmoid_f32(logits), gt, n, priors, 1000.0)
    assert np.array_equal(ml.match.cpu().numpy(), ref["match"])
    l2 = loss2.cpu().numpy().copy()
    assert np.isclose(l2[0], ref["loc_loss"], rtol=1e-5) and np.isclose(l2[1], ref["conf_loss"], rtol=1e-5)
    rdl, rdz = R.add_loss_grads(raw, logits, gt, priors, 1000.0, ref["match"])
    dl_full, dz_full = dl.cpu().numpy().copy(), dz.cpu().numpy().copy()
    assert np.allclose(dl_full, rdl, rtol=1e-4, atol=1e-6) and np.allclose(dz_full, rdz, rtol=1e-4, atol=1e-6)
    # additivity over images: four quarter batches
    q = B // 4
    mq = L.MultiboxLoss(priors, q, G, 1000.0)
    tot = np.zeros(2, np.float64)
    for i in range(4):
        sl = slice(i * q, (i + 1) * q)
        l2q, dlq, dzq = mq.forward_backward(torch.from_numpy(raw[sl]).cuda(), torch.from_numpy(logits[sl]).cuda(),
                                            torch.from_numpy(gt[sl]).cuda(), torch.from_numpy(n[sl]).cuda())
        torch.cuda.synchronize()
        tot += l2q.cpu().numpy()
        assert np.array_equal(dlq.cpu().numpy(), dl_full[sl]) and np.array_equal(dzq.cpu().numpy(), dz_full[sl])
    assert np.allclose(tot, l2, rtol=1e-5)


def test_add_loss_reference_api_known_answers(torch_cuda):
    """model_tests.py:104-263 on the loss: signs, exact zero location loss with no gt."""
    torch = torch_cuda
    from multibox_amd import loss as L
    from oracle import ref_numpy as R
    rng = np.random.RandomState(0)
    P = 646
    priors = rng.uniform(0, 1, (P, 4)).astype(np.float32)
    raw = (rng.randn(2, P, 4) * 0.1).astype(np.float32)
    confs = R.sigmoid_f32(rng.randn(2, P, 1))
    gt = np.zeros((2, 5, 4), np.float32)
    gt[0, 0] = [.1, .1, .9, .9]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for sl, n in [(slice(0, 1), [1]), (slice(0, 1), [0]), (slice(0, 2), [1, 0])]:
        loc, conf = L.add_loss(T(raw[sl]), T(confs[sl]), T(gt[sl]), T(np.array(n, np.int32)), priors, 1.0)
        ref = R.add_loss(raw[sl], confs[sl].reshape(len(n), P), gt[sl], n, priors, 1.0)
        assert np.isclose(float(loc), ref["loc_loss"], rtol=1e-5) and np.isclose(float(conf), ref["conf_loss"], rtol=1e-5)
        if sum(n) == 0:
            assert float(loc) == 0.0
        else:
            assert float(loc) > 0
        assert float(conf) > 0


def _meta_from_golden(g):
    from multibox_amd import detect as D
    return D.make_patch_meta(g["meta_offset"], g["meta_dims"], g["meta_flipped"], g["meta_restrictions"],
                             g["meta_max_to_keep"], g["meta_image_hw"])


@pytest.mark.parametrize("k", ["k5", "k7"])
def test_decode_filter_topk_golden(torch_cuda, golden, k):
    torch = torch_cuda
    from multibox_amd import detect as D
    g = golden.detect
    priors = golden.priors[k + "_restrict"].astype(np.float32)
    raw, confs = g[k + "_raw"], g[k + "_confs"]
    B, P = raw.shape[:2]
    pp = D.DetectPostprocess(priors, B, k_max=200)
    boxes, scores, index, count = pp(torch.from_numpy(raw).cuda(), torch.from_numpy(confs.reshape(B, P)).cuda(), _meta_from_golden(g))
    boxes, scores, count = boxes.cpu().numpy(), scores.cpu().numpy(), count.cpu().numpy()
    assert np.array_equal(count, g[k + "_counts"])
    for b in range(B):
        eb, es = g["%s_b%d_boxes" % (k, b)], g["%s_b%d_scores" % (k, b)]
        assert scores[b, :count[b]].tobytes() == es.tobytes()
        if len(np.unique(es)) == len(es):
            assert boxes[b, :count[b]].tobytes() == eb.tobytes()
        else:
            assert sorted(map(tuple, boxes[b, :count[b]])) == sorted(map(tuple, eb))


def test_decode_filter_topk_full_size_properties(torch_cuda, golden):
    """BASELINE config 4: B=256 patches, k=7 -> P=904, whole-image restrictions, max_to_keep 200."""
    torch = torch_cuda
    from multibox_amd import detect as D
    from oracle import ref_numpy as R
    priors = golden.priors["k7_restrict"].astype(np.float32)
    B, P = 256, priors.shape[0]
    rng = np.random.RandomState(5)
    raw = (rng.randn(B, P, 4) * 0.05).astype(np.float32)
    confs = R.sigmoid_f32(rng.randn(B, P) * 2)
    confs[0, :300] = 0.5                                     # a big tie group
    offs = np.zeros((B, 2), np.int32)
    dims = np.tile([[480, 640]], (B, 1))
    flips = (np.arange(B) % 2).astype(np.int32)
    res = np.tile([[0, 0, 1, 1]], (B, 1)).astype(np.float32)
    mtk = np.full((B,), 200, np.int32)
    mtk[3] = 0
    meta = D.make_patch_meta(offs, dims, flips, res, mtk, dims)
    pp = D.DetectPostprocess(priors, B, k_max=200)
    boxes, scores, index, count = [t.cpu().numpy() for t in pp(torch.from_numpy(raw).cuda(), torch.from_numpy(confs).cuda(), meta)]
    assert count[3] == 0 and (np.delete(count, 3) == 200).all()
    for b in [0, 1, 2, 100, 255]:
        rb, rs, ridx = R.detect_postprocess(raw[b], confs[b], priors, res[b], mtk[b], offs[b], dims[b], dims[b], flips[b])
        assert np.array_equal(index[b, :count[b]], ridx)      # same tie rule as the oracle
        assert scores[b, :count[b]].tobytes() == rs.tobytes()
        assert boxes[b, :count[b]].tobytes() == rb.tobytes()
        assert (np.diff(scores[b, :count[b]]) <= 0).all()     # sortedness


def test_optional_nms_matches_oracle_and_properties():
    """mbx_nms (row N1, optional, off by default: the reference has no NMS): keep decisions equal the numpy restatement
    exactly (float64 IoU, same operation order), survivors keep their order, no two survivors overlap above the
    threshold, threshold 1.0 keeps everything, ragged counts and empty patches are honoured."""
    import torch
    import __graft_entry__ as g
    g.build()
    from multibox_amd import _lib
    from oracle import ref_numpy as R
    l = _lib.lib()
    rng = np.random.RandomState(5)
    B, K = 9, 200
    xy = rng.uniform(0, .8, (B, K, 2)); wh = rng.uniform(.02, .3, (B, K, 2))
    boxes = np.concatenate([xy, xy + wh], -1)
    boxes[1, 50:60] = boxes[1, 40:50]                                    # exact duplicates
    boxes[2, :, :] = boxes[2, :1, :]                                     # one box 200 times
    boxes[3, :, 2:] = boxes[3, :, :2]                                    # zero-area boxes: IoU 0 by definition
    scores = np.sort(rng.uniform(0, 1, (B, K)).astype(np.float32), axis=1)[:, ::-1].copy()
    index = np.tile(np.arange(K, dtype=np.int32), (B, 1))
    count = np.array([200, 200, 200, 200, 0, 1, 63, 64, 129], np.int32)
    for thr in (0.5, 0.3, 1.0, 0.0):
        tb, ts, ti, tc = [torch.from_numpy(a.copy()).cuda() for a in (boxes, scores, index, count)]
        _lib.check(l.mbx_nms(tb.data_ptr(), ts.data_ptr(), ti.data_ptr(), tc.data_ptr(), B, K, float(thr),
                             torch.cuda.current_stream().cuda_stream), "mbx_nms")
        ob, os_, oi, oc = [t.cpu().numpy() for t in (tb, ts, ti, tc)]
        for b in range(B):
            keep = R.nms_greedy(boxes[b, :count[b]], thr)
            assert oc[b] == len(keep), (thr, b, oc[b], len(keep))
            assert np.array_equal(oi[b, :oc[b]], keep.astype(np.int32)), (thr, b)
            assert ob[b, :oc[b]].tobytes() == boxes[b][keep].tobytes() and os_[b, :oc[b]].tobytes() == scores[b][keep].tobytes()
            if thr == 1.0:
                assert oc[b] == count[b]
            if count[b] > 0:
                assert oi[b, 0] == 0                                     # the best box always survives
        assert oc[2] == (1 if thr < 1.0 else 200) and oc[3] == 200 and oc[4] == 0 and oc[5] == 1


def test_detect_postprocess_with_nms_option():
    """DetectPostprocess(nms_iou=...) = the reference path followed by the NMS stage; nms_iou=None is the reference."""
    import torch
    import __graft_entry__ as g
    g.build()
    from multibox_amd import detect as D
    from oracle import ref_numpy as R
    rng = np.random.RandomState(8)
    P, B = 646, 4
    priors = rng.uniform(0.1, 0.5, (P, 4)).astype(np.float32)
    priors[:, 2:] = priors[:, :2] + rng.uniform(0.05, 0.4, (P, 2)).astype(np.float32)
    raw = (rng.randn(B, P, 4) * 0.02).astype(np.float32)
    confs = rng.uniform(0, 1, (B, P)).astype(np.float32)
    dims = np.tile([[299, 299]], (B, 1))
    meta = D.make_patch_meta(np.zeros((B, 2), np.int32), dims, np.zeros((B,), np.int32),
                             np.tile([[0, 0, 1, 1]], (B, 1)).astype(np.float32), np.full((B,), 200, np.int32), dims)
    T = lambda a: torch.from_numpy(a).cuda()
    base = [t.cpu().numpy().copy() for t in D.DetectPostprocess(priors, B, k_max=200)(T(raw), T(confs), meta)]
    out = [t.cpu().numpy() for t in D.DetectPostprocess(priors, B, k_max=200, nms_iou=0.45)(T(raw), T(confs), meta)]
    for b in range(B):
        keep = R.nms_greedy(base[0][b, :base[3][b]], 0.45)
        assert out[3][b] == len(keep) < base[3][b]
        assert np.array_equal(out[2][b, :len(keep)], base[2][b][keep]) and out[0][b, :len(keep)].tobytes() == base[0][b][keep].tobytes()
