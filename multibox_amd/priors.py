"""Prior (default box) generation -- host mirror of the reference's priors.py.

``generate_priors`` has the reference's signature and return type
(priors.py:185: a Python list of ``[x1, y1, x2, y2]``) and is computed by the C-ABI
``mbx_generate_priors`` (float64, bit-exact).  ``priors.pkl`` keeps the reference's
format (README.md:10-18: a pickle of that list).
"""
from __future__ import annotations

import pickle

import numpy as np

from . import _lib

DEFAULT_GRIDS = (8, 6, 4, 3, 2, 1)   # priors.py:196


def head_grids(input_size=299):
    """Grid sizes of the detection heads for a given INPUT_SIZE (model.py:198-293).

    The backbone emits f x f features (8 at 299, 14 at 512); the heads emit
    f, f-2, ceil(f/2), ceil(f/2)-1, ceil(f/2)-2 cells and one (f-7)^2 'global' head.
    At 299 this is the reference's hard-coded [8,6,4,3,2,1]; other sizes are the
    build's generalisation, returned as (grids, cells_of_last_head).
    """
    s = input_size
    s = (s - 3) // 2 + 1          # Conv2d_1a 3x3 s2 VALID
    s = s - 2                     # Conv2d_2a 3x3 VALID
    s = (s - 3) // 2 + 1          # MaxPool_3a
    s = s - 2                     # Conv2d_4a 3x3 VALID
    s = (s - 3) // 2 + 1          # MaxPool_5a
    s = (s - 3) // 2 + 1          # Mixed_6a
    f = (s - 3) // 2 + 1          # Mixed_7a
    h = (f + 1) // 2
    return [f, f - 2, h, h - 1, h - 2], (f - 7) ** 2


def generate_priors(aspect_ratios, min_scale=0.1, max_scale=0.95, restrict_to_image_bounds=True, grids=None):
    """priors.py:185-314 through libmbx.  Returns list of [x1,y1,x2,y2] Python floats."""
    return generate_priors_array(aspect_ratios, min_scale, max_scale, restrict_to_image_bounds, grids).tolist()


def priors_for_input_size(aspect_ratios, input_size=299, min_scale=0.1, max_scale=0.95, restrict_to_image_bounds=True):
    """Priors matching the network's prediction order at any INPUT_SIZE (float64 [P,4]).

    At 299 this is generate_priors (priors.py:185-314: grids 8,6,4,3,2 + one whole-image box).  At other
    sizes the last head is a g x g map with ONE prior per cell (7x7 at 512): the gridded part
    keeps the reference's scale ladder, the last head's cells get aspect-ratio-1 boxes at max_scale.
    """
    grids, last = head_grids(input_size)
    if last == 1:
        return generate_priors_array(aspect_ratios, min_scale, max_scale, restrict_to_image_bounds, grids + [1])
    g = int(round(last ** 0.5))
    main = generate_priors_array(aspect_ratios, min_scale, max_scale, restrict_to_image_bounds, grids + [1])[:-1]
    tail = generate_priors_array([1.0], max_scale, max_scale, restrict_to_image_bounds, [g, 1])[:-1]
    return np.concatenate([main, tail])


def generate_priors_array(aspect_ratios, min_scale=0.1, max_scale=0.95, restrict_to_image_bounds=True, grids=None):
    l = _lib.lib()
    ars = np.ascontiguousarray(aspect_ratios, dtype=np.float64)
    g = np.ascontiguousarray(DEFAULT_GRIDS if grids is None else grids, dtype=np.int32)
    n = l.mbx_priors_count(len(ars), g.ctypes.data, len(g))
    if n < 0:
        _lib.check(n, "mbx_priors_count")
    out = np.zeros((n, 4), np.float64)
    _lib.check(l.mbx_generate_priors(ars.ctypes.data, len(ars), float(min_scale), float(max_scale),
                                     int(bool(restrict_to_image_bounds)), g.ctypes.data, len(g), out.ctypes.data),
               "mbx_generate_priors")
    return out


def save_priors(path, priors):
    """README.md:16-17 format: pickle of list-of-lists (plain floats; protocol 2 so py2 can read it)."""
    with open(path, "wb") as f:
        pickle.dump([[float(v) for v in row] for row in priors], f, protocol=2)


def load_priors(path):
    """train.py:368-370 / detect.py:511-513: accepts py2 text-mode pickles with numpy scalars."""
    with open(path, "rb") as f:
        p = pickle.load(f, encoding="latin1")
    return np.array(p).astype(np.float32)


def generate_aspect_ratios(dataset, num_aspect_ratios=11, visualize=False, warp_bboxes=True, random_state=None):
    """Dataset-specific aspect ratios by 1-D k-means (priors.py:11-183) -- host code, offline.

    dataset: list of dicts with 'id', 'width', 'height', 'object' -> 'bbox' -> 'xmin/xmax/ymin/ymax' (normalised
    coordinates), as the reference's dataset functions return.  Same arithmetic as the reference: boxes warped so
    that the image is square (the SHORTER side's coordinates are stretched, priors.py:41-48), aspect = w / h,
    kept if |a - 5| <= 1e-8 + 5*5 (``np.isclose(a, 5, 5)`` -- rtol 5, priors.py:62) and finite, KMeans, centres
    sorted by membership count, largest first (priors.py:90-93).  ``visualize`` (matplotlib windows and
    ``raw_input`` prompts in the reference) is accepted and ignored.  The reference does not seed KMeans;
    ``random_state`` is this build's addition.  Returns a numpy array of ``num_aspect_ratios`` floats.
    """
    from collections import Counter
    from sklearn.cluster import KMeans
    feats = []
    for image_data in dataset:
        bb = image_data["object"]["bbox"]
        xmin = np.atleast_2d(np.asarray(bb["xmin"], dtype=np.float64)).T.copy()
        xmax = np.atleast_2d(np.asarray(bb["xmax"], dtype=np.float64)).T.copy()
        ymin = np.atleast_2d(np.asarray(bb["ymin"], dtype=np.float64)).T.copy()
        ymax = np.atleast_2d(np.asarray(bb["ymax"], dtype=np.float64)).T.copy()
        if warp_bboxes:
            w, h = float(image_data["width"]), float(image_data["height"])
            if w > h:
                ymin *= w / h
                ymax *= w / h
            else:
                xmin *= h / w
                xmax *= h / w
        with np.errstate(divide="ignore", invalid="ignore"):
            feats.extend(((xmax - xmin) / (ymax - ymin)).tolist())
    X = np.array(feats, dtype=np.float64).reshape(-1, 1)
    X = X[np.isclose(X[:, 0], 5, 5)]
    X = X[~np.isinf(X).any(axis=1)]
    cluster = KMeans(n_clusters=num_aspect_ratios, n_init=10, random_state=random_state)
    cluster.fit(X)
    cnt = Counter(list(cluster.labels_.ravel()))
    cf = [[float(f[0]), cnt[i]] for i, f in enumerate(cluster.cluster_centers_)]
    cf.sort(key=lambda x: x[1])          # stable sort + reverse, as the reference does (tie order included)
    cf.reverse()
    return np.array([x[0] for x in cf])
