"""COCO bounding-box AP / AR -- the metric step of the reference's eval.py:212-246.

The reference hands its top-100 boxes per image to pycocotools' COCOeval (iouType 'bbox', useCats = 0) and logs the
twelve summary numbers.  pycocotools is a third-party dependency that is absent here, so this restates its published
algorithm (cocoeval.py: evaluateImg / accumulate / summarize) for the case the reference uses: one category, no
crowd annotations.  "Parity unpinned": there is no pycocotools in this image to check against; the tests pin the
behaviour on hand-computed cases.

evaluate_bbox(gt, dt) with
  gt: list of {"image_id", "bbox": [x, y, w, h], "area", ["iscrowd": 0]}          (eval.py:176-186)
  dt: array-like rows [image_id, x, y, w, h, score, category]                      (eval.py:166-173)
returns the 12 statistics in COCOeval.stats order and the summary lines in its print format.
"""
from __future__ import annotations

import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, 10)
REC_THRS = np.linspace(0.0, 1.0, 101)
AREA_RNG = [(0.0, 1e10), (0.0, 32.0 ** 2), (32.0 ** 2, 96.0 ** 2), (96.0 ** 2, 1e10)]
AREA_LBL = ["all", "small", "medium", "large"]
MAX_DETS = [1, 10, 100]


def _iou_xywh(d, g):
    """IoU matrix [len(d), len(g)] of xywh boxes (maskUtils.iou with iscrowd = 0)."""
    if len(d) == 0 or len(g) == 0:
        return np.zeros((len(d), len(g)))
    dx0, dy0, dx1, dy1 = d[:, 0:1], d[:, 1:2], d[:, 0:1] + d[:, 2:3], d[:, 1:2] + d[:, 3:4]
    gx0, gy0, gx1, gy1 = g[:, 0], g[:, 1], g[:, 0] + g[:, 2], g[:, 1] + g[:, 3]
    iw = np.clip(np.minimum(dx1, gx1) - np.maximum(dx0, gx0), 0, None)
    ih = np.clip(np.minimum(dy1, gy1) - np.maximum(dy0, gy0), 0, None)
    inter = iw * ih
    union = d[:, 2:3] * d[:, 3:4] + (g[:, 2] * g[:, 3])[None] - inter
    return np.where(union > 0, inter / np.where(union > 0, union, 1), 0.0)


def _evaluate_img(dt, gt, a_rng, max_det):
    """cocoeval.evaluateImg: dt rows [x,y,w,h,score] (any order), gt rows [x,y,w,h,area].  Returns
    (scores, matched [T,D] bool, det_ignore [T,D] bool, number of non-ignored gt)."""
    g_ig = ~((gt[:, 4] >= a_rng[0]) & (gt[:, 4] <= a_rng[1])) if len(gt) else np.zeros(0, bool)
    g_order = np.argsort(g_ig, kind="mergesort")                     # non-ignored first
    gt, g_ig = gt[g_order], g_ig[g_order]
    d_order = np.argsort(-dt[:, 4], kind="mergesort")[:max_det] if len(dt) else np.zeros(0, int)
    dt = dt[d_order]
    ious = _iou_xywh(dt[:, :4], gt[:, :4])
    T, D, G = len(IOU_THRS), len(dt), len(gt)
    dtm = -np.ones((T, D), int)
    gtm = -np.ones((T, G), int)
    dt_ig = np.zeros((T, D), bool)
    for ti, t in enumerate(IOU_THRS):
        for di in range(D):
            iou, m = min(t, 1 - 1e-10), -1
            for gi in range(G):
                if gtm[ti, gi] >= 0:
                    continue                                          # already matched (no crowd gt here)
                if m > -1 and not g_ig[m] and g_ig[gi]:
                    break                                             # matched to a regular gt: stop at the ignored ones
                if ious[di, gi] < iou:
                    continue
                iou, m = ious[di, gi], gi
            if m == -1:
                continue
            dt_ig[ti, di] = g_ig[m]
            dtm[ti, di] = m
            gtm[ti, m] = di
    d_area = dt[:, 2] * dt[:, 3]
    out_of_range = (d_area < a_rng[0]) | (d_area > a_rng[1])
    dt_ig = dt_ig | ((dtm == -1) & out_of_range[None, :])
    return dt[:, 4], dtm >= 0, dt_ig, int((~g_ig).sum())


def evaluate_bbox(gt_annotations, pred_annotations):
    pred = np.asarray(pred_annotations, dtype=np.float64).reshape(-1, 7) if len(pred_annotations) else np.zeros((0, 7))
    gt_by_img, dt_by_img = {}, {}
    for a in gt_annotations:
        x, y, w, h = a["bbox"]
        gt_by_img.setdefault(a["image_id"], []).append([x, y, w, h, a.get("area", w * h)])
    for r in pred:
        dt_by_img.setdefault(int(r[0]), []).append([r[1], r[2], r[3], r[4], r[5]])
    img_ids = sorted(set(gt_by_img) | set(dt_by_img))
    T, R, A, M = len(IOU_THRS), len(REC_THRS), len(AREA_RNG), len(MAX_DETS)
    precision = -np.ones((T, R, A, M))
    recall = -np.ones((T, A, M))
    for ai, a_rng in enumerate(AREA_RNG):
        per_img = []
        for i in img_ids:
            g = np.asarray(gt_by_img.get(i, []), np.float64).reshape(-1, 5)
            d = np.asarray(dt_by_img.get(i, []), np.float64).reshape(-1, 5)
            if len(g) == 0 and len(d) == 0:
                continue
            per_img.append(_evaluate_img(d, g, a_rng, MAX_DETS[-1]))
        for mi, max_det in enumerate(MAX_DETS):
            if not per_img:
                continue
            scores = np.concatenate([e[0][:max_det] for e in per_img])
            order = np.argsort(-scores, kind="mergesort")
            dtm = np.concatenate([e[1][:, :max_det] for e in per_img], axis=1)[:, order]
            dig = np.concatenate([e[2][:, :max_det] for e in per_img], axis=1)[:, order]
            npig = sum(e[3] for e in per_img)
            if npig == 0:
                continue
            tps = np.cumsum(dtm & ~dig, axis=1).astype(np.float64)
            fps = np.cumsum(~dtm & ~dig, axis=1).astype(np.float64)
            for ti in range(T):
                tp, fp = tps[ti], fps[ti]
                nd = len(tp)
                rc = tp / npig
                pr = tp / (fp + tp + np.spacing(1))
                recall[ti, ai, mi] = rc[-1] if nd else 0
                pr = pr.tolist()
                for k in range(nd - 1, 0, -1):
                    if pr[k] > pr[k - 1]:
                        pr[k - 1] = pr[k]
                q = np.zeros(R)
                inds = np.searchsorted(rc, REC_THRS, side="left")
                for ri, pi in enumerate(inds):
                    if pi < nd:
                        q[ri] = pr[pi]
                precision[ti, :, ai, mi] = q

    def summarize(ap, iou_thr=None, area="all", max_det=100):
        ai, mi = AREA_LBL.index(area), MAX_DETS.index(max_det)
        s = precision[:, :, ai, mi] if ap else recall[:, ai, mi]
        if iou_thr is not None:
            s = s[np.where(np.isclose(IOU_THRS, iou_thr))[0]]
        v = float(np.mean(s[s > -1])) if (s > -1).any() else -1.0
        title, typ = ("Average Precision", "(AP)") if ap else ("Average Recall", "(AR)")
        iou_s = "%0.2f:%0.2f" % (IOU_THRS[0], IOU_THRS[-1]) if iou_thr is None else "%0.2f" % iou_thr
        return v, " %-18s %s @[ IoU=%-9s | area=%6s | maxDets=%3d ] = %0.3f" % (title, typ, iou_s, area, max_det, v)
    spec = [(1, None, "all", 100), (1, .5, "all", 100), (1, .75, "all", 100), (1, None, "small", 100), (1, None, "medium", 100),
            (1, None, "large", 100), (0, None, "all", 1), (0, None, "all", 10), (0, None, "all", 100), (0, None, "small", 100),
            (0, None, "medium", 100), (0, None, "large", 100)]
    out = [summarize(*sp) for sp in spec]
    return [v for v, _ in out], [line for _, line in out]
