"""Row F1: the independent restatement oracle/ref_inputs.py -- (1) pinned by hand-computed known answers,
(2) the PRODUCT's host path (multibox_amd/inputs.py) held against it.  The HIP path is held against it in
tests/test_gpu_augment.py.  TF 0.11's own pixels do not exist in this image: parity with TF itself stays unpinned."""
import numpy as np
import pytest

from oracle import ref_inputs as O
from multibox_amd import inputs as I


def test_resize_known_answers():
    rng = np.random.RandomState(5)
    img = rng.rand(8, 12, 3).astype(np.float32)
    const = np.full((7, 9, 3), 0.25, np.float32)
    for m in range(4):
        assert np.allclose(O.RESIZE[m](const, 5, 11), 0.25, atol=1e-6)                 # a flat picture stays flat
        assert np.allclose(O.RESIZE[m](img, 8, 12), img, atol=1e-6)                    # identity size
        assert O.RESIZE[m](img, 5, 7).shape == (5, 7, 3)
    # bilinear without half-pixel centres: 2 x 3 -> 4 x 6 samples the source at 0, .5, 1, 1.5, 2, 2.5 (clamped)
    a = np.arange(6, dtype=np.float32).reshape(2, 3, 1)
    y = O.resize_bilinear(a, 4, 6)[:, :, 0]
    assert np.allclose(y[0], [0, .5, 1, 1.5, 2, 2]) and np.allclose(y[1], [1.5, 2, 2.5, 3, 3.5, 3.5]) and np.allclose(y[3], y[2])
    assert np.array_equal(O.resize_nearest(img, 4, 6), img[::2, ::2])                  # src = floor(dst * 2)
    assert np.allclose(O.resize_area(img, 4, 6), img.reshape(4, 2, 6, 2, 3).mean((1, 3)), atol=1e-6)   # 2 x 2 box means
    # area with a fractional scale, by hand: 3 -> 2 pixels covers [0, 1.5) and [1.5, 3): weights (1, .5) and (.5, 1) / 1.5
    row = np.array([[[0.0], [0.6], [0.9]]], np.float32).repeat(3, 2)
    got = O.resize_area(row, 1, 2)[0, :, 0]
    assert np.allclose(got, [(0.0 + 0.5 * 0.6) / 1.5, (0.5 * 0.6 + 0.9) / 1.5], atol=1e-6)
    # cubic convolution reproduces a linear ramp away from the borders
    ramp = np.tile((np.arange(16, dtype=np.float32) / 16.0)[None, :, None], (4, 1, 3))
    up = O.resize_bicubic(ramp, 4, 32)
    assert np.allclose(up[:, 4:-6, 0], (np.arange(32) * 0.5 / 16.0)[4:-6][None], atol=1e-5)
    # Keys weights (A = -0.75) at t = 0 and t = .5
    assert np.allclose(O._keys(0.0), [0, 1, 0, 0]) and np.allclose(O._keys(0.5), [-0.09375, 0.59375, 0.59375, -0.09375])


def test_colour_known_answers():
    px = np.array([[[0.2, 0.4, 0.6]]], np.float32)
    # HSV of (0.2, 0.4, 0.6): v = 0.6, s = 0.4 / 0.6, h = (4 + (0.2 - 0.4) / 0.4) / 6 = 3.5 / 6
    h, s, v = O.rgb_to_hsv(px[0, 0])
    assert np.allclose([h, s, v], [3.5 / 6.0, 2.0 / 3.0, 0.6], atol=1e-7)
    assert np.allclose(O.hsv_to_rgb(h, s, v), px[0, 0], atol=1e-7)
    assert np.allclose(O.distort_color(px, [(O.BRIGHTNESS, 0.1)]), px + 0.1, atol=1e-7)
    assert np.allclose(O.distort_color(px, [(O.BRIGHTNESS, 0.9)]), [[[1.0, 1.0, 1.0]]])            # the final clip
    assert np.allclose(O.distort_color(px, [(O.SATURATION, 0.0)]), [[[0.6, 0.6, 0.6]]], atol=1e-7)  # s = 0: grey at v
    assert np.allclose(O.distort_color(px, [(O.SATURATION, 1.5)]), [[[0.0, 0.3, 0.6]]], atol=1e-7)  # s -> 1 (clipped)
    # hue + 1/3 rotates the channels: (r, g, b) -> (b, r, g)
    assert np.allclose(O.distort_color(px, [(O.HUE, 1.0 / 3.0)]), [[[0.6, 0.2, 0.4]]], atol=1e-6)
    two = np.array([[[0.2, 0.2, 0.2], [0.6, 0.6, 0.6]]], np.float32)
    assert np.allclose(O.distort_color(two, [(O.CONTRAST, 0.5)]), [[[0.3] * 3, [0.5] * 3]], atol=1e-7)   # about the mean 0.4
    # the orders of distort_color (inputs.py:71-91) do not commute: contrast before / after brightness clipping differ
    img = np.random.RandomState(2).rand(5, 7, 3).astype(np.float32)
    args = {O.BRIGHTNESS: 0.1, O.SATURATION: 1.3, O.HUE: -0.1, O.CONTRAST: 0.7}
    outs = [O.distort_color(img, [(op, args[op]) for op in O.ORDERS[k]]) for k in range(4)]
    assert all(o.min() >= 0 and o.max() <= 1 for o in outs)
    assert not np.allclose(outs[0], outs[1]) and not np.allclose(outs[2], outs[3])


def test_box_arithmetic_known_answers():
    """inputs.py:128-180 and :184-203."""
    # crop (y=50, x=100, h=100, w=200) of a 200 x 400 image
    xmin, ymin = np.array([.30, .10, .00], np.float32), np.array([.30, .10, .00], np.float32)
    xmax, ymax = np.array([.60, .20, .26], np.float32), np.array([.60, .20, .26], np.float32)
    cx0, cy0, cx1, cy1 = O.crop_boxes(xmin, ymin, xmax, ymax, 200, 400, (50, 100, 100, 200), minimum_area=50)
    # box 0: px (120..240, 60..120) -> clipped to the crop and shifted: x 20..140, y 10..70 -> /200, /100
    # box 1: px (40..80, 20..40): entirely left/above the crop -> zero area -> dropped
    # box 2: px (0..104, 0..52): 4 x 2 px inside the crop = 8 px^2 <= 50 -> dropped
    assert len(cx0) == 1
    assert np.allclose([cx0[0], cy0[0], cx1[0], cy1[0]], [20 / 200., 10 / 100., 140 / 200., 70 / 100.], atol=1e-6)
    # the reference clips to the IMAGE size, not the crop size (:153-156): a box overhanging the crop keeps its overhang
    bx = O.crop_boxes(np.array([.25], np.float32), np.array([.25], np.float32), np.array([1.], np.float32),
                      np.array([1.], np.float32), 200, 400, (50, 100, 100, 200), 50)
    assert np.allclose([v[0] for v in bx], [0.0, 0.0, 1.0, 1.0])
    sx0, sy0, sx1, sy1 = O.shift_boxes([.5, .0], [.5, .01], [.6, 1.], [.7, .99], 200, 400,
                                       ([.01, .01], [.01, .01], [.02, .02], [.02, .02]))
    assert np.allclose(sx0, [.49, 0]) and np.allclose(sx1, [.61, 1]) and np.allclose(sy0, [.48, 0]) and np.allclose(sy1, [.72, 1])
    fx0, fx1 = O.flip_boxes([.1, .5], [.3, .9])
    assert np.allclose(fx0, [.7, .1]) and np.allclose(fx1, [.9, .5])


# ------------------------------------------------------------------ the product's host path against the restatement
@pytest.mark.parametrize("shape,out", [((37, 53), 24), ((20, 20), 31), ((64, 9), 16)])
def test_product_resize_matches_the_restatement(shape, out):
    img = np.random.RandomState(shape[0]).rand(shape[0], shape[1], 3).astype(np.float32)
    for m in range(4):
        got, want = I.RESIZE_METHODS[m](img, out, out), O.RESIZE[m](img, out, out)
        if m in (O.BILINEAR, O.NEAREST):
            assert np.array_equal(got, want), m          # TF's kernel order in float32 IS the definition: bit for bit
        else:
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-6, err_msg="method %d" % m)


def test_product_colour_ops_match_the_restatement():
    rng = np.random.RandomState(9)
    img = rng.rand(11, 13, 3).astype(np.float32)
    img[0, 0] = 0.5                                      # a grey pixel: hue undefined, saturation 0
    for fast in (True, False):
        for ordering in ((0, 1) if fast else (0, 1, 2, 3)):
            ops = I.color_ops(ordering, fast, rng)
            order = (O.FAST_ORDERS if fast else O.ORDERS)[min(ordering, 1) if fast else ordering]
            assert tuple(op for op, _ in ops) == order    # the application order of inputs.py:65-91
            np.testing.assert_allclose(I.apply_color_ops(img, ops), O.distort_color(img, ops), rtol=0, atol=1e-6)
    assert (I.COLOR_BRIGHTNESS, I.COLOR_SATURATION, I.COLOR_HUE, I.COLOR_CONTRAST) == (O.BRIGHTNESS, O.SATURATION, O.HUE, O.CONTRAST)


def test_product_box_arithmetic_matches_the_restatement():
    rng = np.random.RandomState(4)
    for _ in range(50):
        n = rng.randint(0, 6)
        x0 = rng.uniform(0, .8, n).astype(np.float32); y0 = rng.uniform(0, .8, n).astype(np.float32)
        x1 = (x0 + rng.uniform(.01, .2, n)).astype(np.float32); y1 = (y0 + rng.uniform(.01, .2, n)).astype(np.float32)
        H, W = int(rng.randint(100, 400)), int(rng.randint(100, 400))
        ch, cw = int(rng.randint(20, H)), int(rng.randint(20, W))
        crop = (int(rng.randint(0, H - ch + 1)), int(rng.randint(0, W - cw + 1)), ch, cw)
        got = I.crop_bboxes(x0, y0, x1, y1, H, W, crop, 50)
        want = O.crop_boxes(x0, y0, x1, y1, H, W, crop, 50)
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
        # the shift: replay the product's four draws (x_min, x_max, y_min, y_max order, inputs.py:194-197)
        seed = int(rng.randint(1 << 30))
        got = I.shift_bboxes(x0, y0, x1, y1, H, W, 5, np.random.RandomState(seed))
        r = np.random.RandomState(seed)
        mw, mh = np.float32(1.0 / W * 5), np.float32(1.0 / H * 5)
        dxm, dxM = r.uniform(0, mw, n).astype(np.float32), r.uniform(0, mw, n).astype(np.float32)
        dym, dyM = r.uniform(0, mh, n).astype(np.float32), r.uniform(0, mh, n).astype(np.float32)
        want = O.shift_boxes(x0, y0, x1, y1, H, W, (dxm, dxM, dym, dyM))
        assert all(np.array_equal(g, w) for g, w in zip(got, want))


def test_product_apply_plan_matches_the_restatement():
    rng = np.random.RandomState(12)
    img = rng.rand(40, 56, 3).astype(np.float32)
    for method in range(4):
        p = I.AugmentPlan()
        p.crop, p.method, p.flip = (3, 7, 30, 41), method, bool(method & 1)
        p.color = I.color_ops(method, False, rng)
        want = O.augment_pixels(img, p.crop, p.method, p.color, p.flip, 24)
        np.testing.assert_allclose(I.apply_plan(img, p, 24), want, rtol=0, atol=2e-6)
