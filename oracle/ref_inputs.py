"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the pixel and box arithmetic of the reference's training input graph
(row F1 of SURVEY 8f), independent of the product modules (multibox_amd/inputs.py is the product's host path and
csrc/augment.hip its device path: both are CHECKED against this file, neither imports it).

What the reference does (/root/reference/inputs.py):
  :184-203  distorted_shifted_bounding_box -- boxes grow by up to N pixels, clipped to [0, 1]
  :100-181  distorted_bounding_box_crop    -- crop box from tf.image.sample_distorted_bounding_box, then the box arithmetic
  :296-302  tf.image.resize_images with a randomly drawn ResizeMethod (0 bilinear, 1 nearest, 2 bicubic, 3 area)
  :44-98    distort_color                  -- brightness / saturation / hue / contrast in one of four orders, clip to [0, 1]
  :323-327  random flip, :350-351 (x - 0.5) * 2
The arithmetic of tf.image.* lives in tensorflow==0.11.0rc0 (requirements.txt:7), which is NOT under /root/reference and
cannot be installed here: the functions below restate TF 0.11's published kernels (resize_bilinear_op.cc,
resize_nearest_neighbor_op.cc, resize_bicubic_op.cc with A = -0.75, resize_area_op.cc, adjust_* in image_ops.py,
rgb_to_hsv / hsv_to_rgb in colorspace_op.h) -- all without half-pixel centres, src = dst * in / out.  PARITY UNPINNED
against TF itself (no TF-produced pixels exist in this image); pinned here are hand-computed known answers
(tests/test_inputs_cpu.py) and the box arithmetic, which is plain and follows the reference line by line.
Random DRAWS are not restated: every function takes the drawn values as arguments.

Everything is written per output pixel / per box in the obvious way (float64 unless TF's kernel order in float32 is
the definition); sizes in the tests are small.
"""
from __future__ import annotations

import math

import numpy as np

BILINEAR, NEAREST, BICUBIC, AREA = 0, 1, 2, 3          # tf.image.ResizeMethod


# ----------------------------------------------------------------------------------------------- resize
def _scale(n_in, n_out):
    return np.float32(np.float32(n_in) / np.float32(n_out))     # TF: in / static_cast<float>(out), float32


def resize_bilinear(img, out_h, out_w):
    """resize_bilinear_op.cc (TF 0.11, align_corners=False), float32 in the kernel's order:
    top = tl + (tr - tl) * x_lerp; bottom = bl + (br - bl) * x_lerp; out = top + (bottom - top) * y_lerp."""
    img = np.asarray(img, np.float32)
    H, W, C = img.shape
    hs, ws = _scale(H, out_h), _scale(W, out_w)
    out = np.empty((out_h, out_w, C), np.float32)
    for y in range(out_h):
        in_y = np.float32(y) * hs
        y0 = int(math.floor(in_y)); y1 = min(int(math.ceil(in_y)), H - 1); yl = np.float32(in_y - np.float32(y0))
        for x in range(out_w):
            in_x = np.float32(x) * ws
            x0 = int(math.floor(in_x)); x1 = min(int(math.ceil(in_x)), W - 1); xl = np.float32(in_x - np.float32(x0))
            top = img[y0, x0] + (img[y0, x1] - img[y0, x0]) * xl
            bot = img[y1, x0] + (img[y1, x1] - img[y1, x0]) * xl
            out[y, x] = top + (bot - top) * yl
    return out


def resize_nearest(img, out_h, out_w):
    """resize_nearest_neighbor_op.cc (TF 0.11): src = min(floor(dst * scale), in - 1)."""
    img = np.asarray(img, np.float32)
    H, W, _ = img.shape
    hs, ws = _scale(H, out_h), _scale(W, out_w)
    ys = [min(int(math.floor(np.float32(y) * hs)), H - 1) for y in range(out_h)]
    xs = [min(int(math.floor(np.float32(x) * ws)), W - 1) for x in range(out_w)]
    return np.stack([np.stack([img[y, x] for x in xs], 0) for y in ys], 0)


def _keys(t, a=-0.75):
    """Keys' cubic convolution kernel weights for the taps at -1, 0, +1, +2 around a sample at fraction t (float64).
    (TF 0.11 reads them from a 1024-entry table of this function; the closed form differs from it by < 1e-3.)"""
    def near(x):      # |x| <= 1
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0

    def far(x):       # 1 < |x| < 2
        return ((a * x - 5.0 * a) * x + 8.0 * a) * x - 4.0 * a
    return [far(t + 1.0), near(t), near(1.0 - t), far(2.0 - t)]


def resize_bicubic(img, out_h, out_w):
    """resize_bicubic_op.cc (TF 0.11): 4 x 4 taps, indices clamped to the image, rows first then columns, float64."""
    img = np.asarray(img, np.float32).astype(np.float64)
    H, W, C = img.shape
    hs, ws = _scale(H, out_h), _scale(W, out_w)
    rows = np.empty((out_h, W, C))
    for y in range(out_h):
        in_y = np.float32(y) * hs
        y0 = int(math.floor(in_y)); w = _keys(float(np.float32(in_y - np.float32(y0))))
        rows[y] = sum(w[k] * img[min(max(y0 - 1 + k, 0), H - 1)] for k in range(4))
    out = np.empty((out_h, out_w, C))
    for x in range(out_w):
        in_x = np.float32(x) * ws
        x0 = int(math.floor(in_x)); w = _keys(float(np.float32(in_x - np.float32(x0))))
        out[:, x] = sum(w[k] * rows[:, min(max(x0 - 1 + k, 0), W - 1)] for k in range(4))
    return out.astype(np.float32)


def resize_area(img, out_h, out_w):
    """resize_area_op.cc: the mean of the source rectangle [dst * scale, (dst + 1) * scale) in both axes, border pixels
    weighted by the covered fraction, source indices clamped to the image.  Float64, no particular order."""
    img = np.asarray(img, np.float32).astype(np.float64)
    H, W, C = img.shape
    hs, ws = H / float(out_h), W / float(out_w)

    def taps(o, scale, n):
        lo, hi = o * scale, (o + 1) * scale
        res = []
        for i in range(int(math.floor(lo)), int(math.ceil(hi))):
            cover = min(hi, i + 1.0) - max(lo, float(i))
            if cover > 0:
                res.append((min(i, n - 1), cover))
        tot = sum(c for _, c in res)
        return [(i, c / tot) for i, c in res]
    out = np.zeros((out_h, out_w, C))
    xt = [taps(x, ws, W) for x in range(out_w)]
    for y in range(out_h):
        for iy, wy in taps(y, hs, H):
            for x in range(out_w):
                for ix, wx in xt[x]:
                    out[y, x] += wy * wx * img[iy, ix]
    return out.astype(np.float32)


RESIZE = {BILINEAR: resize_bilinear, NEAREST: resize_nearest, BICUBIC: resize_bicubic, AREA: resize_area}


# ----------------------------------------------------------------------------------------------- colour
BRIGHTNESS, SATURATION, HUE, CONTRAST = 0, 1, 2, 3
# distort_color's four full orders (inputs.py:71-91) and the two fast ones (:65-70)
ORDERS = {0: (BRIGHTNESS, SATURATION, HUE, CONTRAST), 1: (SATURATION, BRIGHTNESS, CONTRAST, HUE),
          2: (CONTRAST, HUE, BRIGHTNESS, SATURATION), 3: (HUE, SATURATION, CONTRAST, BRIGHTNESS)}
FAST_ORDERS = {0: (BRIGHTNESS, SATURATION), 1: (SATURATION, BRIGHTNESS)}


def rgb_to_hsv(px):
    """colorspace_op.h RGBToHSV on one pixel (float64): v = max, s = range / max, h in [0, 1)."""
    r, g, b = (float(c) for c in px)
    v, lo = max(r, g, b), min(r, g, b)
    rng = v - lo
    s = rng / v if v > 0 else 0.0
    if rng <= 0:
        return 0.0, s, v
    if v == r:
        h = (g - b) / rng
    elif v == g:
        h = 2.0 + (b - r) / rng
    else:
        h = 4.0 + (r - g) / rng
    return (h / 6.0) % 1.0, s, v


def hsv_to_rgb(h, s, v):
    """colorspace_op.h HSVToRGB: the piecewise-linear channel ramps of TF's kernel."""
    dh = h * 6.0
    dr = min(max(abs(dh - 3.0) - 1.0, 0.0), 1.0)
    dg = min(max(2.0 - abs(dh - 2.0), 0.0), 1.0)
    db = min(max(2.0 - abs(dh - 4.0), 0.0), 1.0)
    return ((1.0 - s + s * dr) * v, (1.0 - s + s * dg) * v, (1.0 - s + s * db) * v)


def distort_color(img, ops):
    """inputs.py:44-98 with the drawn arguments given: ops = [(BRIGHTNESS, delta) | (SATURATION, factor) | (HUE, delta) |
    (CONTRAST, factor)] in application order; tf.image.adjust_* of TF 0.11 in float64; the final clip to [0, 1]."""
    x = np.asarray(img, np.float32).astype(np.float64).copy()
    H, W, _ = x.shape
    for op, arg in ops:
        if op == BRIGHTNESS:                                  # adjust_brightness: x + delta
            x = x + arg
        elif op == CONTRAST:                                  # adjust_contrast: (x - mean_c) * factor + mean_c, mean over H, W
            for c in range(3):
                m = x[:, :, c].sum() / (H * W)
                x[:, :, c] = (x[:, :, c] - m) * arg + m
        else:
            for i in range(H):
                for j in range(W):
                    h, s, v = rgb_to_hsv(x[i, j])
                    if op == SATURATION:                      # adjust_saturation: s * factor clipped to [0, 1]
                        s = min(max(s * arg, 0.0), 1.0)
                    else:                                     # adjust_hue: (h + delta) mod 1
                        h = (h + arg) % 1.0
                    x[i, j] = hsv_to_rgb(h, s, v)
    return np.clip(x, 0.0, 1.0).astype(np.float32)


# ----------------------------------------------------------------------------------------------- boxes
def shift_boxes(xmin, ymin, xmax, ymax, image_height, image_width, draws):
    """inputs.py:184-203.  draws = (dx_min, dx_max, dy_min, dy_max): the four uniform vectors in the reference's order,
    each already scaled to [0, max_shift / size); boxes grow outwards and are clipped to [0, 1]."""
    dxm, dxM, dym, dyM = (np.asarray(d, np.float32) for d in draws)
    f = np.float32
    return (np.clip(f(xmin) - dxm, 0, 1).astype(f), np.clip(f(ymin) - dym, 0, 1).astype(f),
            np.clip(f(xmax) + dxM, 0, 1).astype(f), np.clip(f(ymax) + dyM, 0, 1).astype(f))


def crop_boxes(xmin, ymin, xmax, ymax, image_height, image_width, crop, minimum_area):
    """inputs.py:128-180, box by box.  crop = (y, x, h, w) in pixels (bbox_begin, bbox_size).  Boxes are scaled by the
    RECORD's image size, intersected with the crop, moved to its origin, clipped to [0, image size] (the reference clips
    to the image size, not the crop size: :153-156), dropped when the area is <= minimum_area (:158-159) and divided by
    the crop size (:175-178).  float32 like the TF graph."""
    f = np.float32
    y0, x0, ch, cw = (f(v) for v in crop)
    H, W = f(image_height), f(image_width)
    out = [[], [], [], []]
    for bx0, by0, bx1, by1 in zip(xmin, ymin, xmax, ymax):
        sy0 = min(max(max(f(by0) * H, y0) - y0, f(0)), H)
        sx0 = min(max(max(f(bx0) * W, x0) - x0, f(0)), W)
        sy1 = min(max(min(f(by1) * H, y0 + ch) - y0, f(0)), H)
        sx1 = min(max(min(f(bx1) * W, x0 + cw) - x0, f(0)), W)
        if f(sx1 - sx0) * f(sy1 - sy0) > f(minimum_area):
            for lst, v in zip(out, (f(sx0) / cw, f(sy0) / ch, f(sx1) / cw, f(sy1) / ch)):
                lst.append(f(v))
    return tuple(np.asarray(l, np.float32) for l in out)


def flip_boxes(xmin, xmax):
    """inputs.py:323-327: xmin' = 1 - xmax, xmax' = 1 - xmin."""
    return np.float32(1.0) - np.asarray(xmax, np.float32), np.float32(1.0) - np.asarray(xmin, np.float32)


# ----------------------------------------------------------------------------------------------- one example
def augment_pixels(image01, crop, method, color_ops, flip, size, scale_to_pm1=False):
    """The pixel half of inputs.py:272-351 for drawn decisions: slice the crop (y, x, h, w) or None, resize to size x size
    with `method`, colour ops, flip, optionally (x - 0.5) * 2."""
    img = np.asarray(image01, np.float32)
    if crop is not None:
        y, x, h, w = (int(v) for v in crop)
        img = img[y:y + h, x:x + w]
    img = RESIZE[method](img, size, size)
    if color_ops:
        img = distort_color(img, color_ops)
    if flip:
        img = img[:, ::-1]
    if scale_to_pm1:
        img = (img - np.float32(0.5)) * np.float32(2.0)
    return np.ascontiguousarray(img, np.float32)
