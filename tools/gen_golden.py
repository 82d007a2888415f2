#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING the reference's own pure-numpy functions.

Runs only in the build container (needs /root/reference); the fixtures it writes
are data (inputs + expected outputs) and travel with the repo.  Nothing from the
reference source is written anywhere: each file is read, converted py2->py3 IN
MEMORY with lib2to3, the wanted FunctionDefs are selected with ``ast`` and
exec'd with numpy / scipy in the namespace.

The one semantic patch: loss.py:16 ``locations.shape[0] / batch_size``
is Python-2 integer division; the Div node is rewritten to FloorDiv.  Everything
else is the reference's code, on scipy 1.15.3 / numpy 2.2 (the reference pins
scipy 0.17 / numpy 1.11: same optimum, possibly different choice among exact
ties, so generated cases avoid ties).
"""
import ast
import hashlib
import json
import os
import sys

import numpy as np
import scipy.optimize

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _load(fname, wanted, extra_ns=None, patch=None):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
    src = open(os.path.join(REF, fname)).read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    src3 = str(tool.refactor_string(src + "\n", fname))
    tree = ast.parse(src3)
    body = [n for n in tree.body
            if (isinstance(n, ast.FunctionDef) and n.name in wanted)
            or (isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id in wanted for t in n.targets))]
    mod = ast.Module(body=body, type_ignores=[])
    if patch:
        mod = patch(mod)
    ast.fix_missing_locations(mod)
    ns = {"np": np, "linear_sum_assignment": scipy.optimize.linear_sum_assignment}
    ns.update(extra_ns or {})
    exec(compile(mod, fname, "exec"), ns)
    return ns


class _FloorDivNumPred(ast.NodeTransformer):
    """loss.py:16 only: ``num_predictions = locations.shape[0] / batch_size``."""

    def visit_Assign(self, node):
        if (len(node.targets) == 1 and isinstance(node.targets[0], ast.Name)
                and node.targets[0].id == "num_predictions"
                and isinstance(node.value, ast.BinOp) and isinstance(node.value.op, ast.Div)):
            node.value.op = ast.FloorDiv()
        return node


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest = {}

    # ---------------------------------------------------------------- priors (A1)
    pr = _load("priors.py", {"generate_priors"})
    gp = pr["generate_priors"]
    ar5 = [1.0, 2.0, 3.0, 1.0 / 2.0, 1.0 / 3.0]
    ar7 = [1.0, 2.0, 3.0, 1.0 / 2.0, 1.0 / 3.0, 1.5, 1.0 / 1.5]
    ar8 = ar7 + [4.0]
    pri = {}
    for name, ars, kw in [("k5_restrict", ar5, {}), ("k5_clip", ar5, {"restrict_to_image_bounds": False}),
                          ("k7_restrict", ar7, {}), ("k8_restrict", ar8, {}),
                          ("k5_scales", ar5, {"min_scale": 0.2, "max_scale": 0.8})]:
        p = np.array(gp(ars, **kw), dtype=np.float64)
        pri[name] = p
        pri[name + "_ars"] = np.array(ars, np.float64)
        manifest["priors_" + name] = dict(rows=int(p.shape[0]),
                                          sha256_f8=hashlib.sha256(p.astype("<f8").tobytes()).hexdigest(),
                                          sha256_f4=hashlib.sha256(p.astype("<f4").tobytes()).hexdigest())
    np.savez_compressed(os.path.join(OUT, "priors.npz"), **pri)

    # --------------------------------------------------------------- matching (A6)
    ls = _load("loss.py", {"compute_assignments", "SMALL_EPSILON"},
               patch=lambda m: _FloorDivNumPred().visit(m))
    ca = ls["compute_assignments"]
    p5 = pri["k5_restrict"].astype(np.float32)
    p7 = pri["k7_restrict"].astype(np.float32)

    def synth(seed, B, P, G, priors, n_list=None, loc_sigma=0.05):
        rng = np.random.RandomState(seed)
        raw = (rng.randn(B, P, 4) * loc_sigma).astype(np.float32)
        logits = (rng.randn(B, P) * 2.0 - 2.0).astype(np.float32)
        confs = (1.0 / (1.0 + np.exp(-logits.astype(np.float64)))).astype(np.float32)
        n = np.array(n_list if n_list is not None else rng.randint(0, G + 1, size=B), np.int32)
        gt = np.zeros((B, G, 4), np.float32)
        for b in range(B):
            xy = rng.uniform(0, 0.7, size=(n[b], 2))
            wh = rng.uniform(0.05, 0.3, size=(n[b], 2))
            gt[b, :n[b], :2] = xy
            gt[b, :n[b], 2:] = xy + wh
        return raw, logits, confs, gt, n

    cases = {}
    specs = [("b2_p646_g5", 0, 2, 5, p5, [3, 0]),
             ("b4_p646_g13", 1, 4, 13, p5, [13, 0, 1, 7]),
             ("b3_p904_g100", 2, 3, 100, p7, [100, 37, 0]),
             ("b2_p646_g13_wide", 3, 2, 13, p5, [13, 12]),
             ("b8_p646_g13_rand", 4, 8, 13, p5, None)]
    for name, seed, B, G, priors, n_list in specs:
        raw, logits, confs, gt, n = synth(seed, B, priors.shape[0], G, priors, n_list,
                                          loc_sigma=0.3 if "wide" in name else 0.05)
        P = priors.shape[0]
        dec = (raw + priors[None]).astype(np.float32).reshape(-1, 4)        # loss.py:71
        c = (confs.reshape(-1) + np.float32(1e-10)).astype(np.float32)      # loss.py:74
        part, stacked = ca(dec.copy(), c.copy(), gt.copy(), n.copy(), np.int32(B), np.float32(1000.0))
        cases[name + "_raw"] = raw
        cases[name + "_logits"] = logits
        cases[name + "_confs"] = confs
        cases[name + "_gt"] = gt
        cases[name + "_n"] = n
        cases[name + "_part"] = np.asarray(part, np.int32)
        cases[name + "_stacked"] = np.asarray(stacked, np.float32)
        manifest["match_" + name] = dict(B=B, P=int(P), G=G, alpha=1000.0, n=[int(x) for x in n],
                                         matched_rows=[int(x) for x in np.nonzero(part)[0]][:40])
    # edge: confidences saturated at exactly 0 and 1 (loss.py:22-25 clamps)
    raw, logits, confs, gt, n = synth(7, 2, 646, 5, p5, [5, 2])
    confs[0, :10] = 1.0
    confs[0, 10:20] = 0.0
    confs[1, 5] = 1.0
    dec = (raw + p5[None]).astype(np.float32).reshape(-1, 4)
    c = (confs.reshape(-1) + np.float32(1e-10)).astype(np.float32)
    part, stacked = ca(dec.copy(), c.copy(), gt.copy(), n.copy(), np.int32(2), np.float32(1.0))
    for k, v in dict(raw=raw, logits=logits, confs=confs, gt=gt, n=n, part=np.asarray(part, np.int32),
                     stacked=np.asarray(stacked, np.float32)).items():
        cases["sat_alpha1_" + k] = v
    manifest["match_sat_alpha1"] = dict(B=2, P=646, G=5, alpha=1.0, n=[5, 2])
    np.savez_compressed(os.path.join(OUT, "matching.npz"), **cases)

    # ---------------------------------------------------- detect post-process (A9-13)
    dt = _load("detect.py", {"extract_patches", "filter_proposals", "convert_proposals"})
    fp, cp, ep = dt["filter_proposals"], dt["convert_proposals"], dt["extract_patches"]
    det = {}
    rng = np.random.RandomState(11)
    metas = [  # offset(y,x), dims(h,w), flipped, restrictions, max_to_keep, image(h,w)
        ((0, 0), (480, 640), 0, [0, 0, 1, 1], 200, (480, 640)),
        ((0, 0), (480, 640), 1, [0, 0, 1, 1], 100, (480, 640)),
        ((113, 226), (299, 299), 0, [0.1, 0.1, 0.9, 1.0], 50, (480, 640)),
        ((0, 339), (299, 299), 1, [0.1, 0.0, 1.0, 0.9], 50, (480, 640)),
        ((69, 138), (185, 185), 0, [0.1, 0.1, 0.9, 0.9], 50, (375, 500)),
        ((0, 0), (333, 500), 0, [0.45, 0.45, 0.55, 0.55], 200, (333, 500)),   # filters (almost) everything
    ]
    for k, priors in (("k5", p5), ("k7", p7)):
        P = priors.shape[0]
        B = len(metas)
        raw = (rng.randn(B, P, 4) * 0.08).astype(np.float32)
        confs = (1.0 / (1.0 + np.exp(-(rng.randn(B, P, 1) * 2.0)))).astype(np.float32)
        det[k + "_raw"] = raw
        det[k + "_confs"] = confs
        counts = []
        for b, (off, dims, flip, res, mtk, imhw) in enumerate(metas):
            boxes = np.clip(raw[b] + priors, 0.0, 1.0)                       # detect.py:412-413
            fb, fc = fp(boxes, confs[b], np.array(res, np.float32))          # detect.py:416
            if fb.shape[0] == 0:                                             # detect.py:419-420
                counts.append(0)
                det["%s_b%d_boxes" % (k, b)] = np.zeros((0, 4))
                det["%s_b%d_scores" % (k, b)] = np.zeros((0,), np.float32)
                continue
            order = np.argsort(fc.ravel())[::-1][:mtk]                       # detect.py:423-424
            fb, fc = fb[order], fc[order]
            cb = cp(fb, np.array(off, np.int32), np.array(dims, np.int32), np.array(imhw, np.int32), flip)
            counts.append(int(cb.shape[0]))
            det["%s_b%d_boxes" % (k, b)] = np.asarray(cb, np.float64)
            det["%s_b%d_scores" % (k, b)] = np.asarray(fc, np.float32).ravel()
        det[k + "_counts"] = np.array(counts, np.int32)
    det["meta_offset"] = np.array([m[0] for m in metas], np.int32)
    det["meta_dims"] = np.array([m[1] for m in metas], np.int32)
    det["meta_flipped"] = np.array([m[2] for m in metas], np.int32)
    det["meta_restrictions"] = np.array([m[3] for m in metas], np.float32)
    det["meta_max_to_keep"] = np.array([m[4] for m in metas], np.int32)
    det["meta_image_hw"] = np.array([m[5] for m in metas], np.int32)
    # extract_patches offsets / restrictions (A13)
    for tag, hw, pd, st in [("480x640_299_113", (480, 640), (299, 299), (113, 113)),
                            ("375x500_185_69", (375, 500), (185, 185), (69, 69)),
                            ("200x200_299_113", (200, 200), (299, 299), (113, 113))]:
        img = np.zeros(hw + (3,), np.float32)
        patches, offs, res, cnt = ep(img, pd, st)
        det["patches_%s_offsets" % tag] = np.asarray(offs, np.int32)
        det["patches_%s_restrictions" % tag] = np.asarray(res, np.float32)
        manifest["patches_" + tag] = int(cnt)
    np.savez_compressed(os.path.join(OUT, "detect.npz"), **det)

    manifest["_generated_with"] = dict(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0],
                                       patch="loss.py:16 Div->FloorDiv (py2 integer division)")
    # ------------------------------------------------ aspect-ratio clustering (F4, priors.py:11-183)
    # The reference's KMeans is unseeded; well separated clusters make the centres unique up to float noise.
    from sklearn.cluster import KMeans as _KM

    def _KMeansCompat(n_clusters=8, n_jobs=None, **kw):   # sklearn >= 1.0 dropped n_jobs (the reference passes n_jobs=8)
        return _KM(n_clusters=n_clusters, n_init=10, **kw)

    class _Plt:                            # visualize=False: never called
        def __getattr__(self, k):
            raise RuntimeError("matplotlib is not part of the golden run")

    from collections import Counter
    ar_ns = _load("priors.py", {"generate_aspect_ratios"}, extra_ns={"KMeans": _KMeansCompat, "plt": _Plt(), "Counter": Counter})
    rng = np.random.RandomState(7)
    centres = [0.5, 1.0, 2.0, 3.5]
    sizes = [40, 90, 60, 25]
    dataset = []
    img = 0
    for c, n in zip(centres, sizes):
        for j in range(n):
            W_, H_ = [(640, 480), (480, 640), (500, 500)][(img + j) % 3]
            a = c * (1.0 + 0.01 * rng.randn())             # aspect of the box in the SQUARE-warped image
            hh = 0.2 + 0.1 * rng.rand()
            ww = a * hh
            s_ = max(W_, H_) / float(min(W_, H_))
            # un-warp: the reference stretches the shorter side's coordinates by s_
            if W_ > H_:
                bw, bh = ww, hh / s_
            else:
                bw, bh = ww / s_, hh
            x0, y0 = 0.05 + 0.1 * rng.rand(), 0.05 + 0.1 * rng.rand()
            dataset.append({"id": img, "width": W_, "height": H_, "filename": "",
                            "object": {"bbox": {"xmin": [x0], "xmax": [x0 + bw], "ymin": [y0], "ymax": [y0 + bh]}}})
            img += 1
    # two degenerate boxes: zero height (inf aspect, dropped at priors.py:67) and aspect 40 (dropped at priors.py:62)
    dataset.append({"id": img, "width": 500, "height": 500, "filename": "",
                    "object": {"bbox": {"xmin": [0.1], "xmax": [0.3], "ymin": [0.2], "ymax": [0.2]}}})
    dataset.append({"id": img + 1, "width": 500, "height": 500, "filename": "",
                    "object": {"bbox": {"xmin": [0.1], "xmax": [0.9], "ymin": [0.2], "ymax": [0.22]}}})
    import copy, io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        out_w = ar_ns["generate_aspect_ratios"](copy.deepcopy(dataset), num_aspect_ratios=4, visualize=False, warp_bboxes=True)
        out_n = ar_ns["generate_aspect_ratios"](copy.deepcopy(dataset), num_aspect_ratios=4, visualize=False, warp_bboxes=False)
    np.savez(os.path.join(OUT, "aspect_ratios.npz"),
             width=np.array([d["width"] for d in dataset]), height=np.array([d["height"] for d in dataset]),
             bbox=np.array([[d["object"]["bbox"][k][0] for k in ("xmin", "ymin", "xmax", "ymax")] for d in dataset]),
             expected_warp=np.asarray(out_w, np.float64).ravel(), expected_nowarp=np.asarray(out_n, np.float64).ravel())
    manifest["aspect_ratios"] = dict(boxes=len(dataset), warp=[float(v) for v in np.asarray(out_w).ravel()],
                                     nowarp=[float(v) for v in np.asarray(out_n).ravel()])

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
