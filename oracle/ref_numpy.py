"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's host-side hot path.

Every function cites the reference file:line it restates.  The arithmetic ORDER and
dtypes of the reference are kept on purpose (float32 cost pieces, float64 cost
matrix, float64 prior maths), because the parity bar is bit-exact for priors,
match indices and the decode path.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against the
fixtures in ``tests/golden/`` which ``tools/gen_golden.py`` produced by running
the reference's own functions (loaded from /root/reference in the build
container; see that script for the one documented py2->py3 patch).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.optimize import linear_sum_assignment

SMALL_EPSILON = 1e-10  # loss.py:6

DEFAULT_GRIDS = (8, 6, 4, 3, 2, 1)  # priors.py:196


# --------------------------------------------------------------------------- A1
def _one_prior(cx, cy, scale, a, restrict):
    """One box; priors.py:264-310 (and the 1x1 special case 206-258)."""
    ra = np.sqrt(a)
    w = scale * ra
    h = scale / ra
    x1 = cx - (w / 2.0)
    x2 = cx + (w / 2.0)
    y1 = cy - (h / 2.0)
    y2 = cy + (h / 2.0)
    if restrict:
        # priors.py:277-303.  NB the reference names are swapped (right/left) but
        # only the max is used.
        wt = max(abs(min(0, x1)), abs(min(0, 1 - x2)))
        ht = max(abs(min(0, y1)), abs(min(0, 1 - y2)))
        trim = max(wt, ht)
        if h > w:
            wt, ht = trim * a, trim
        else:
            wt, ht = trim, trim / a
        xa, xb = x1 + wt, x2 - wt
        ya, yb = y1 + ht, y2 - ht
        x1, x2 = min(xa, xb), max(xa, xb)
        y1, y2 = min(ya, yb), max(ya, yb)
    return [max(x1, 0.0), max(y1, 0.0), min(x2, 1.0), min(y2, 1.0)]


def generate_priors(aspect_ratios, min_scale=0.1, max_scale=0.95,
                    restrict_to_image_bounds=True, grids=DEFAULT_GRIDS):
    """priors.py:185-314.  Returns float64 [P,4]; order grid-major, row i, col j, aspect.

    ``grids`` generalises the hard-coded list (priors.py:196) for SURVEY D4; a grid
    of 1 gets the single a=1 box like the reference's 1x1 head.
    """
    grids = list(grids)
    ns = len(grids)
    out = []
    for idx, g in enumerate(grids):
        i1 = idx + 1
        scale = min_scale + (max_scale - min_scale) * (i1 - 1) / (ns - 1) if ns > 1 else min_scale
        if g == 1:
            out.append(_one_prior(0.5, 0.5, scale, 1.0, restrict_to_image_bounds))
            continue
        for i in range(g):
            for j in range(g):
                cy = (i + 0.5) / g
                cx = (j + 0.5) / g
                for a in aspect_ratios:
                    out.append(_one_prior(cx, cy, scale, a, restrict_to_image_bounds))
    return np.array(out, dtype=np.float64).reshape(-1, 4)


# --------------------------------------------------------------------------- A6
def cost_matrix(locations, confidences, gt, alpha):
    """Cost C[P, n] of one image; loss.py:21-35.  float32 pieces, float64 result.

    locations [P,4] f32 (prior-decoded), confidences [P] f32 (already +1e-10),
    gt [n,4] f32.
    """
    locations = np.asarray(locations, np.float32)
    confidences = np.asarray(confidences, np.float32)
    gt = np.asarray(gt, np.float32).reshape(-1, 4)
    lc = np.log(confidences)                        # loss.py:21
    v = np.float32(1.0) - confidences               # loss.py:22
    v[v > 1.0] = 1.0
    v[v <= 0] = SMALL_EPSILON
    l1c = np.log(v)                                 # loss.py:25
    half_alpha = np.float32(np.float32(alpha) / np.float32(2.0))
    P = locations.shape[0]
    C = np.zeros((P, gt.shape[0]), dtype=np.float64)
    for j in range(gt.shape[0]):
        d = locations - gt[j]                       # f32
        s = d * d                                   # np.linalg.norm: (x.conj()*x).real
        ss = ((s[:, 0] + s[:, 1]) + s[:, 2]) + s[:, 3]   # add.reduce over 4 contiguous f32
        nrm = np.sqrt(ss)                           # f32
        C[:, j] = (half_alpha * (nrm ** 2) - lc) + l1c   # loss.py:35, left to right, f32
    return C


def compute_assignments(locations, confidences, gt_bboxes, num_gt_bboxes, batch_size, alpha):
    """loss.py:8-53.  Returns (partition int32 [B*P], stacked_gt f32 [M,4], match int32 [B,P]).

    ``match[b, p]`` = gt index matched to prediction p of image b, or -1 -- the
    representation the C-ABI ``mbx_match`` returns; the reference's two outputs are
    derived from it exactly as loss.py:44-48 builds them (ascending row order).
    """
    locations = np.asarray(locations, np.float32).reshape(-1, 4)
    confidences = np.asarray(confidences, np.float32).reshape(-1)
    gt_bboxes = np.asarray(gt_bboxes, np.float32)
    B = int(batch_size)
    P = locations.shape[0] // B
    part = np.zeros(B * P, dtype=np.int32)
    match = np.full((B, P), -1, dtype=np.int32)
    stacked = []
    for b in range(B):
        n = int(num_gt_bboxes[b])
        if n == 0:
            continue
        sl = slice(b * P, (b + 1) * P)
        C = cost_matrix(locations[sl], confidences[sl], gt_bboxes[b, :n], alpha)
        rows, cols = linear_sum_assignment(C)       # loss.py:40
        for r, c in zip(rows, cols):                # rows ascending
            part[b * P + r] = 1
            match[b, r] = c
            stacked.append(gt_bboxes[b, c])
    stacked = (np.array(stacked, dtype=np.float32).reshape(-1, 4)
               if stacked else np.zeros((0, 4), np.float32))
    return part, stacked, match


# --------------------------------------------------------------------------- A7
def sigmoid_f32(z):
    z = np.asarray(z, np.float32)
    return (np.float32(1.0) / (np.float32(1.0) + np.exp(-z))).astype(np.float32)


def add_loss(raw_locs, confs, gt_bboxes, num_gt_bboxes, priors, alpha, match=None):
    """loss.py:55-116 forward.  raw_locs [B,P,4] f32, confs [B,P] f32 sigmoid outputs.

    Returns dict(loc_loss, conf_loss, match, decoded).  Sums are taken in float64
    and rounded once (TF's reduction order is un-vendored; tests use a tolerance).
    """
    raw_locs = np.asarray(raw_locs, np.float32)
    B, P = raw_locs.shape[:2]
    priors = np.asarray(priors, np.float32)
    dec = (raw_locs + priors[None]).astype(np.float32)              # loss.py:71
    c = (np.asarray(confs, np.float32).reshape(B, P) + np.float32(SMALL_EPSILON)).astype(np.float32)  # :74
    if match is None:
        _, _, match = compute_assignments(dec.reshape(-1, 4), c.reshape(-1), gt_bboxes,
                                          num_gt_bboxes, B, alpha)
    gt_bboxes = np.asarray(gt_bboxes, np.float32)
    loc = 0.0
    conf = 0.0
    for b in range(B):
        m = match[b]
        pos = m >= 0
        if pos.any():
            d = (dec[b, pos] - gt_bboxes[b, m[pos]]).astype(np.float32)
            loc += float(np.sum((d * d).astype(np.float64))) / 2.0   # tf.nn.l2_loss
            conf -= float(np.sum(np.log(c[b, pos]).astype(np.float64)))
        neg = ~pos
        u = ((np.float32(1.0) - c[b, neg]) + np.float32(SMALL_EPSILON)).astype(np.float32)
        conf -= float(np.sum(np.log(u).astype(np.float64)))
    return dict(loc_loss=np.float32(alpha) * np.float32(loc), conf_loss=np.float32(conf),
                match=match, decoded=dec)


def add_loss_grads(raw_locs, logits, gt_bboxes, priors, alpha, match):
    """Analytic d(loc_loss+conf_loss)/d raw_locs and /d logits (confs = sigmoid(logits)).

    Follows loss.py:71-101 through the sigmoid of model.py:322; no gradient through
    the matching (py_func, loss.py:82).  float64 maths, returned as float32.
    """
    raw_locs = np.asarray(raw_locs, np.float32)
    B, P = raw_locs.shape[:2]
    priors = np.asarray(priors, np.float32)
    dec = (raw_locs + priors[None]).astype(np.float32)
    z = np.asarray(logits, np.float32).reshape(B, P)
    s = sigmoid_f32(z)
    c = (s + np.float32(SMALL_EPSILON)).astype(np.float32)
    gt_bboxes = np.asarray(gt_bboxes, np.float32)
    dl = np.zeros((B, P, 4), np.float64)
    dz = np.zeros((B, P), np.float64)
    s64 = s.astype(np.float64)
    ds = s64 * (1.0 - s64)
    for b in range(B):
        m = match[b]
        pos = m >= 0
        if pos.any():
            dl[b, pos] = float(alpha) * (dec[b, pos].astype(np.float64) - gt_bboxes[b, m[pos]])
            dz[b, pos] = -ds[b, pos] / c[b, pos].astype(np.float64)
        neg = ~pos
        u = ((np.float32(1.0) - c[b, neg]) + np.float32(SMALL_EPSILON)).astype(np.float64)
        dz[b, neg] = ds[b, neg] / u
    return dl.astype(np.float32), dz.astype(np.float32)


# ---------------------------------------------------------------------- A9..A13
def decode_clip(raw_locs, priors):
    """detect.py:412-413 (same as eval.py:146-148); f32."""
    return np.clip(np.asarray(raw_locs, np.float32) + np.asarray(priors, np.float32), 0.0, 1.0)


def filter_mask(boxes, restrictions=None):
    """detect.py:74-104 as a keep-mask (strict inequalities, original order kept)."""
    r = np.asarray([0.1, 0.1, 0.9, 0.9] if restrictions is None else restrictions)
    b = np.asarray(boxes)
    return ~((b[:, 0] < r[0]) | (b[:, 1] < r[1]) | (b[:, 2] > r[2]) | (b[:, 3] > r[3]))


def convert_proposals(boxes, offset, patch_dims, image_dims, is_flipped=0):
    """detect.py:106-131.  f32 boxes widen to float64 by the broadcast with python floats."""
    xs = float(patch_dims[1]) / float(image_dims[1])
    ys = float(patch_dims[0]) / float(image_dims[0])
    xo = float(offset[1]) / float(image_dims[1])
    yo = float(offset[0]) / float(image_dims[0])
    out = np.asarray(boxes) * np.array([xs, ys, xs, ys]) + np.array([xo, yo, xo, yo])
    if is_flipped:
        out = out.copy()
        x1 = 1.0 - out[:, 2]
        x2 = 1.0 - out[:, 0]
        out[:, 0], out[:, 2] = x1, x2
    return out


def detect_postprocess(raw_locs, confs, priors, restrictions, max_to_keep, offset,
                       patch_dims, image_dims, is_flipped):
    """One patch of detect.py:408-436.  Returns (boxes f64 [n,4], scores f32 [n], kept idx).

    Tie order among equal confidences is undefined in the reference (numpy
    quicksort reversed, detect.py:423); this restatement takes a stable
    descending order with the HIGHER original index first among ties, which is
    what the HIP kernel implements; parity tests compare tie groups as sets.
    """
    boxes = decode_clip(raw_locs, priors)
    keep = np.nonzero(filter_mask(boxes, restrictions))[0]
    if keep.size == 0:
        return np.zeros((0, 4)), np.zeros((0,), np.float32), keep
    c = np.asarray(confs, np.float32).reshape(-1)[keep]
    order = np.argsort(c, kind="stable")[::-1][: int(max_to_keep)]
    idx = keep[order]
    return (convert_proposals(boxes[idx], offset, patch_dims, image_dims, is_flipped),
            c[order], idx)


def extract_patch_offsets(image_hw, patch_dims, strides, non_edge_restriction=0.1):
    """Offsets + per-side restrictions of detect.py:20-72 (the image crop itself is F3)."""
    H, W = image_hw
    ph, pw = patch_dims
    sh, sw = strides
    offs, res = [], []
    for h in range(0, H - ph + 1, sh):
        for w in range(0, W - pw + 1, sw):
            offs.append((h, w))
            res.append([0.0 if w == 0 else non_edge_restriction,
                        0.0 if h == 0 else non_edge_restriction,
                        1.0 if w + pw == W else 1.0 - non_edge_restriction,
                        1.0 if h + ph == H else 1.0 - non_edge_restriction])
    return (np.array(offs, np.int32).reshape(-1, 2), np.array(res, np.float32).reshape(-1, 4))


# --------------------------------------------------------------------------- A8
def decay_steps(num_train_examples, batch_size, num_epochs_per_decay):
    """train.py:193-195 with Python-2 integer division."""
    return int((int(num_train_examples) // int(batch_size)) * num_epochs_per_decay)


def learning_rate(step, lr0, dsteps, factor, staircase=True):
    """tf.train.exponential_decay as called at train.py:198-204 (TF un-vendored)."""
    p = step / float(dsteps)
    if staircase:
        p = math.floor(p)
    return np.float32(lr0) * np.float32(factor) ** np.float32(p)


def rmsprop_step(w, g, ms, mom, lr, decay=0.9, momentum=0.0, eps=1.0):
    """tf.train.RMSPropOptimizer dense update (train.py:207-212; TF un-vendored):
    ms = d*ms + (1-d) g^2 ; mom = m*mom + lr*g/sqrt(ms+eps) ; w -= mom.  f32 in place."""
    ms *= np.float32(decay)
    ms += np.float32(1.0 - decay) * g * g
    mom *= np.float32(momentum)
    mom += np.float32(lr) * g / np.sqrt(ms + np.float32(eps))
    w -= mom


def ema_decay(decay, num_updates):
    """tf.train.ExponentialMovingAverage(decay, num_updates) (train.py:253-256)."""
    return min(decay, (1.0 + num_updates) / (10.0 + num_updates))


def nms_greedy(boxes, iou_threshold):
    """Greedy non-maximum suppression over boxes already sorted by descending score (row N1: NOT part of the reference,
    which has no NMS -- detect.py:408-443; the standard algorithm, restated in the operation order of the HIP kernel so
    that the keep decisions can be compared exactly).  boxes [K,4] float64 x1,y1,x2,y2.  Returns the kept indices."""
    b = np.asarray(boxes, np.float64).reshape(-1, 4)
    keep = []
    for i in range(len(b)):
        ok = True
        ai = (b[i, 2] - b[i, 0]) * (b[i, 3] - b[i, 1])
        for j in keep:
            iw = min(b[j, 2], b[i, 2]) - max(b[j, 0], b[i, 0])
            ih = min(b[j, 3], b[i, 3]) - max(b[j, 1], b[i, 1])
            inter = iw * ih if (iw > 0.0 and ih > 0.0) else 0.0
            union = (b[j, 2] - b[j, 0]) * (b[j, 3] - b[j, 1]) + ai - inter
            iou = inter / union if union > 0.0 else 0.0
            if iou > iou_threshold:
                ok = False
                break
        if ok:
            keep.append(i)
    return np.array(keep, np.int64)
