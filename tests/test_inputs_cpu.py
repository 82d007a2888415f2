"""Host input pipelines (F1/F3 first cut): TFRecord framing + Example parsing round trip, the TF-0.11
bilinear resize restatement, the detect patch generator against the reference's extract_patches goldens."""
import io

import numpy as np
import pytest

from multibox_amd import tfrecord as T, inputs as I
from multibox_amd.config import Cfg


def _jpeg(h, w, seed):
    from PIL import Image
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 255, (h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
    img = Image.fromarray(base).resize((w, h), Image.BILINEAR)
    b = io.BytesIO()
    img.save(b, format="JPEG", quality=95)
    return b.getvalue()


def _make_records(path, specs):
    payloads = []
    for i, (h, w, boxes) in enumerate(specs):
        boxes = np.array(boxes, np.float32).reshape(-1, 4)
        payloads.append(T.make_example({
            "image/id": str(1000 + i), "image/encoded": _jpeg(h, w, i), "image/height": [h], "image/width": [w],
            "image/object/bbox/xmin": [float(x) for x in boxes[:, 0]], "image/object/bbox/ymin": [float(x) for x in boxes[:, 1]],
            "image/object/bbox/xmax": [float(x) for x in boxes[:, 2]], "image/object/bbox/ymax": [float(x) for x in boxes[:, 3]],
            "image/object/bbox/count": [len(boxes)],
            "image/object/area": [float((b[2] - b[0]) * w * (b[3] - b[1]) * h) for b in boxes]}))
    T.write_records(path, payloads)


def test_crc32c_known_answers():
    assert T.crc32c(b"123456789") == 0xE3069283          # standard CRC-32C check value
    assert T.crc32c(b"") == 0


def test_tfrecord_roundtrip(tmp_path):
    p = str(tmp_path / "a.tfrecords")
    ex = {"image/id": "42", "image/encoded": b"\x00\x01\xff" * 50, "image/height": [480], "image/width": [-3],
          "image/object/bbox/xmin": [0.1, 0.25], "image/object/bbox/count": [2]}
    T.write_records(p, [T.make_example(ex), T.make_example({"image/id": "7"})])
    recs = [T.parse_example(r) for r in T.read_records(p, verify=True)]
    assert len(recs) == 2 and recs[0]["image/id"] == [b"42"] and recs[0]["image/encoded"] == [b"\x00\x01\xff" * 50]
    assert recs[0]["image/height"] == [480] and recs[0]["image/width"] == [-3] and recs[0]["image/object/bbox/count"] == [2]
    assert np.allclose(recs[0]["image/object/bbox/xmin"], [0.1, 0.25])
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:-2] + b"zz")
    with pytest.raises(IOError):
        list(T.read_records(p, verify=True))


def test_resize_bilinear_tf_semantics():
    x = np.arange(2 * 3 * 1, dtype=np.float32).reshape(2, 3, 1)
    assert np.array_equal(I.resize_bilinear_tf(x, 2, 3), x)                       # identity
    y = I.resize_bilinear_tf(x, 4, 6)[..., 0]
    # align_corners=False, no half-pixel offset: out[i,j] samples in[i*0.5, j*0.5]; the last row/col clamp
    assert np.allclose(y[0], [0, .5, 1, 1.5, 2, 2]) and np.allclose(y[1], [1.5, 2, 2.5, 3, 3.5, 3.5]) and np.allclose(y[3], y[2])
    z = I.resize_bilinear_tf(np.random.RandomState(0).rand(7, 5, 3).astype(np.float32), 3, 2)
    assert z.shape == (3, 2, 3) and z.dtype == np.float32


def _cfg():
    return Cfg(dict(INPUT_SIZE=299, DETECTION=dict(USE_ORIGINAL_IMAGE=True, ORIGINAL_IMAGE_MAX_TO_KEEP=200,
                    USE_FLIPPED_ORIGINAL_IMAGE=True, FLIPPED_IMAGE_MAX_TO_KEEP=100,
                    CROPS=[dict(HEIGHT=299, WIDTH=299, HEIGHT_STRIDE=113, WIDTH_STRIDE=113, FLIP=False, MAX_TO_KEEP=50),
                           dict(HEIGHT=185, WIDTH=185, HEIGHT_STRIDE=69, WIDTH_STRIDE=69, FLIP=True, MAX_TO_KEEP=25)])))


def test_detect_patches_metadata(golden):
    cfg = _cfg()
    img = np.random.RandomState(1).rand(480, 640, 3).astype(np.float32)
    p, o, d, f, r, k = I.detect_patches_for_image(img, (480, 640), cfg)
    g = golden.detect
    n299 = len(g["patches_480x640_299_113_offsets"])
    n185 = len(I.extract_patches(img, (185, 185), (69, 69))[1])
    assert len(p) == 2 + n299 + n185 and all(x.shape == (299, 299, 3) for x in p)
    assert o[0] == (0, 0) and d[0] == (480, 640) and f[:2] == [0, 1] and k[:2] == [200, 100] and r[0] == (0., 0., 1., 1.)
    assert np.array_equal(np.array(o[2:2 + n299]), g["patches_480x640_299_113_offsets"])          # reference extract_patches
    assert np.allclose(np.array(r[2:2 + n299]), g["patches_480x640_299_113_restrictions"])
    assert set(f[2 + n299:]) == {1} and set(k[2 + n299:]) == {25} and set(d[2 + n299:]) == {(185, 185)}
    # the un-cropped 299 patch at (0,0) is the image region itself, shifted to [-1,1]
    assert np.allclose(p[2], (img[:299, :299] - 0.5) * 2.0, atol=1e-6)
    # the flipped original is the mirror of the original patch source
    assert np.allclose(p[1], I.resize_bilinear_tf(((img - 0.5) * 2.0)[:, ::-1], 299, 299))


def test_detect_and_train_batches(tmp_path):
    path = str(tmp_path / "d.tfrecords")
    _make_records(path, [(320, 400, [[.1, .2, .5, .6]]), (300, 300, []), (350, 310, [[.0, .0, 1., 1.], [.2, .2, .4, .9]])])
    cfg = Cfg(dict(INPUT_SIZE=299, DO_RANDOM_FLIP_LEFT_RIGHT=False,
                   DETECTION=dict(USE_ORIGINAL_IMAGE=True, ORIGINAL_IMAGE_MAX_TO_KEEP=200, USE_FLIPPED_ORIGINAL_IMAGE=True,
                                  FLIPPED_IMAGE_MAX_TO_KEEP=100)))
    bs = list(I.detect_batches([path], cfg, batch_size=4))
    assert len(bs) == 1                                        # 6 patches: one full batch, the last 2 dropped like tf.train.batch
    b = bs[0]
    assert b["images"].shape == (4, 299, 299, 3) and b["images"].min() >= -1 and b["images"].max() <= 1
    assert b["image_ids"] == ["1000", "1000", "1001", "1001"] and b["is_flipped"].ravel().tolist() == [0, 1, 0, 1]
    assert b["image_hw"].tolist() == [[320, 400], [320, 400], [300, 300], [300, 300]]
    assert len(list(I.detect_batches([path], cfg, batch_size=4, keep_partial=True))) == 2
    tb = I.train_batches([path], cfg, batch_size=3, max_num_bboxes=5, num_epochs=1)
    images, boxes, nums, ids = next(tb)
    assert images.shape == (3, 299, 299, 3) and boxes.shape == (3, 5, 4) and nums.tolist() == [1, 0, 2] and ids == ["1000", "1001", "1002"]
    assert np.allclose(boxes[0, 0], [.1, .2, .5, .6]) and not boxes[1].any() and np.allclose(boxes[2, 1], [.2, .2, .4, .9])
    # every augmentation of inputs.py:264-327 switched on: runs, stays normalised, is reproducible per seed
    aug = Cfg(dict(INPUT_SIZE=299, DO_RANDOM_FLIP_LEFT_RIGHT=True, DO_RANDOM_BBOX_SHIFT=1.0, RANDOM_BBOX_SHIFT_EXTENT=4,
                   DO_RANDOM_CROP=1.0, RANDOM_CROP_MIN_OBJECT_COVERED=0.7, RANDOM_CROP_ASPECT_RATIO_RANGE=[0.7, 1.4],
                   RANDOM_CROP_AREA_RANGE=[0.5, 1.0], RANDOM_CROP_MAX_ATTEMPTS=100, RANDOM_CROP_MINIMUM_AREA=50,
                   DO_COLOR_DISTORTION=1.0, COLOR_DISTORT_FAST=False))
    runs = [list(I.train_batches([path], aug, 3, 5, num_epochs=2, seed=11, shuffle=True, min_after_dequeue=2)) for _ in range(2)]
    assert len(runs[0]) == 2
    for (im_a, bb_a, n_a, id_a), (im_b, bb_b, n_b, id_b) in zip(*runs):
        assert np.array_equal(im_a, im_b) and np.array_equal(bb_a, bb_b) and id_a == id_b          # same seed, same stream
        assert im_a.shape == (3, 299, 299, 3) and im_a.min() >= -1 and im_a.max() <= 1
        assert bb_a.min() >= 0 and bb_a.max() <= 1 and (n_a <= 2).all()
        for b, n in zip(bb_a, n_a):
            assert (b[:n, 2] >= b[:n, 0]).all() and (b[:n, 3] >= b[:n, 1]).all() and not b[n:].any()
    assert sorted(sum((r[3] for r in runs[0]), [])) == ["1000", "1000", "1001", "1001", "1002", "1002"]


def test_bbox_shift_and_crop_arithmetic():
    """inputs.py:184-203 and 128-180, known answers."""
    rng = np.random.RandomState(0)
    x0, y0, x1, y1 = [np.array(v, np.float32) for v in ([.5, .0], [.5, .01], [.6, 1.], [.7, .99])]
    sx0, sy0, sx1, sy1 = I.shift_bboxes(x0, y0, x1, y1, 200, 400, 5, rng)
    assert (sx0 <= x0).all() and (sx0 >= np.maximum(x0 - 5 / 400., 0) - 1e-7).all()             # outwards, < extent, clipped
    assert (sx1 >= x1).all() and (sx1 <= np.minimum(x1 + 5 / 400., 1) + 1e-7).all()
    assert (sy0 <= y0).all() and (sy0 >= np.maximum(y0 - 5 / 200., 0) - 1e-7).all() and sx0[1] == 0 and sx1[1] == 1
    # crop (y=50, x=100, h=100, w=200) of a 200 x 400 image
    xmin, ymin = np.array([.30, .10, .00], np.float32), np.array([.30, .10, .00], np.float32)
    xmax, ymax = np.array([.60, .20, .26], np.float32), np.array([.60, .20, .26], np.float32)
    cx0, cy0, cx1, cy1 = I.crop_bboxes(xmin, ymin, xmax, ymax, 200, 400, (50, 100, 100, 200), minimum_area=50)
    # box 0: px (120..240, 60..120) -> clipped to the crop and shifted: x 20..140, y 10..70 -> /200, /100
    # box 1: px (40..80, 20..40): entirely left/above the crop -> zero area -> dropped
    # box 2: px (0..104, 0..52): 4 x 2 px inside the crop = 8 px^2 <= 50 -> dropped
    assert len(cx0) == 1
    assert np.allclose([cx0[0], cy0[0], cx1[0], cy1[0]], [20 / 200., 10 / 100., 140 / 200., 70 / 100.], atol=1e-6)


def test_sample_distorted_bounding_box_constraints():
    rng = np.random.RandomState(3)
    H, W = 240, 320
    boxes = np.array([[.2, .3, .6, .7], [.5, .1, .9, .4]], np.float32)            # ymin, xmin, ymax, xmax
    full = 0
    for _ in range(300):
        y, x, h, w = I.sample_distorted_bounding_box(H, W, boxes, 0.7, (0.7, 1.4), (0.5, 1.0), 100, rng)
        assert 0 <= y and 0 <= x and y + h <= H and x + w <= W and h > 0 and w > 0
        if (y, x, h, w) == (0, 0, H, W):
            full += 1
            continue
        assert 0.5 * H * W <= w * h <= 1.0 * H * W
        assert 0.7 - 0.02 <= w / float(h) <= 1.4 + 0.02                           # the ratio is rounded to whole pixels
        cov = []
        for b in boxes:
            bx0, by0, bx1, by1 = int(b[1] * W), int(b[0] * H), int(b[3] * W), int(b[2] * H)
            inter = max(min(bx1, x + w) - max(bx0, x), 0) * max(min(by1, y + h) - max(by0, y), 0)
            cov.append(inter / float((bx1 - bx0) * (by1 - by0)))
        assert max(cov) >= 0.7
    assert full < 30
    # impossible constraints fall back to the whole image; no boxes -> the image itself is the box to cover
    assert I.sample_distorted_bounding_box(H, W, boxes, 0.7, (0.7, 1.4), (2.0, 3.0), 20, rng) == (0, 0, H, W)
    y, x, h, w = I.sample_distorted_bounding_box(H, W, np.zeros((0, 4), np.float32), 0.7, (0.9, 1.1), (0.8, 1.0), 100, rng)
    assert w * h >= 0.7 * H * W


def test_resize_methods_and_colour():
    rng = np.random.RandomState(5)
    img = rng.rand(8, 12, 3).astype(np.float32)
    const = np.full((7, 9, 3), 0.25, np.float32)
    for f in I.RESIZE_METHODS:
        assert np.allclose(f(const, 5, 11), 0.25, atol=1e-6) and f(img, 5, 7).shape == (5, 7, 3)
        assert np.allclose(f(img, 8, 12), img, atol=1e-6)                         # identity size
    assert np.array_equal(I.resize_nearest_tf(img, 4, 6), img[::2, ::2])          # src = floor(dst * 2)
    assert np.allclose(I.resize_area_tf(img, 4, 6), img.reshape(4, 2, 6, 2, 3).mean((1, 3)), atol=1e-6)
    ramp = np.tile((np.arange(16, dtype=np.float32) / 16.0)[None, :, None], (4, 1, 3))
    up = I.resize_bicubic_tf(ramp, 4, 32)                                         # cubic convolution reproduces a linear ramp
    assert np.allclose(up[:, 4:-6, 0], (np.arange(32) * 0.5 / 16.0)[4:-6][None], atol=1e-5)
    # colour: HSV round trip, range, the saturation step keeps grey pixels grey
    assert np.allclose(I._hsv_to_rgb(I._rgb_to_hsv(img.astype(np.float64))), img, atol=1e-6)
    for fast in (True, False):
        for order in ((0,) if fast else (0, 1, 2, 3)):
            out = I.distort_color(img, order, fast, np.random.RandomState(order))
            assert out.shape == img.shape and out.min() >= 0 and out.max() <= 1 and not np.allclose(out, img)
    with pytest.raises(ValueError):
        I.distort_color(img, 4, False, rng)


def test_eval_batches(tmp_path):
    """eval_inputs.py:20-115: no augmentation, areas padded with the boxes, incomplete last batch dropped."""
    path = str(tmp_path / "e.tfrecords")
    _make_records(path, [(320, 400, [[.1, .2, .5, .6]]), (300, 300, []), (350, 310, [[.0, .0, 1., 1.], [.2, .2, .4, .9]])])
    cfg = Cfg(dict(INPUT_SIZE=299))
    bs = list(I.eval_batches([path], cfg, batch_size=2, max_num_bboxes=4))
    assert len(bs) == 1
    images, boxes, nums, areas, ids = bs[0]
    assert images.shape == (2, 299, 299, 3) and boxes.shape == (2, 4, 4) and areas.shape == (2, 4) and nums.tolist() == [1, 0]
    assert np.isclose(areas[0, 0], .4 * 400 * .4 * 320) and not areas[0, 1:].any() and not areas[1].any() and ids == ["1000", "1001"]
    assert np.allclose(images[0], (I.resize_bilinear_tf(I.decode_image(_jpeg(320, 400, 0)), 299, 299) - 0.5) * 2.0)


def test_parallel_train_input_workers(tmp_path):
    """multibox_amd/input_workers.py (the NUM_INPUT_THREADS queue runners of inputs.py:353-371 as worker PROCESSES over
    a shared-memory ring): one worker, shuffle off = the single-process stream bit for bit; three workers cover every
    record exactly once per epoch, in batches, and stop at the end of the epoch; the prefetcher hands the same batches on."""
    from multibox_amd.input_workers import ParallelTrainInput, DevicePrefetcher
    path = str(tmp_path / "t.tfrecords")
    specs = [(300 + 7 * i, 310 + 5 * i, [[.1, .2, .5, .6]] if i % 3 else []) for i in range(10)]
    _make_records(path, specs)
    cfg = Cfg(dict(INPUT_SIZE=299, DO_RANDOM_FLIP_LEFT_RIGHT=True, DO_COLOR_DISTORTION=1.0, COLOR_DISTORT_FAST=True,
                   NUM_INPUT_THREADS=3))
    ref = list(I.train_batches([path], cfg, 2, 5, num_epochs=1, seed=5))
    src = ParallelTrainInput([path], cfg, 2, 5, num_workers=1, num_epochs=1, seed=5, shuffle=False, tmpdir=str(tmp_path))
    got = list(src)
    src.close()
    assert len(got) == len(ref) == 5
    for (a, b, n, ids), (ra, rb, rn, rids) in zip(got, ref):
        assert np.array_equal(a, ra) and np.array_equal(b, rb) and np.array_equal(n, rn) and ids == rids
    # three workers (cfg.NUM_INPUT_THREADS), two epochs, shuffled: every record twice, whole batches only
    src = ParallelTrainInput([path], cfg, 4, 5, num_epochs=2, seed=9, shuffle=True, capacity=6, min_after_dequeue=3,
                             tmpdir=str(tmp_path))
    assert src.n == 3
    pre = DevicePrefetcher(src, 4, 299, 5, device="cpu", depth=2)
    seen = []
    while True:
        try:
            images, bb, n, ids = pre.next()
        except StopIteration:
            break
        assert tuple(images.shape) == (4, 299, 299, 3) and float(images.min()) >= -1 and float(images.max()) <= 1
        assert tuple(bb.shape) == (4, 5, 4) and n.dtype.is_floating_point is False
        for i, image_id in enumerate(ids):
            want = 1 if (int(image_id) - 1000) % 3 else 0
            assert int(n[i]) == want
        seen += ids
    pre.close()
    assert len(seen) == 20 and sorted(seen) == sorted([str(1000 + i) for i in range(10)] * 2)


def test_device_augment_workers_hand_over_the_same_plans(tmp_path):
    """device_augment=True: workers ship cropped uint8 pixels + the draws; finishing those on the host (apply_plan) must
    reproduce the host path's batches bit for bit -- same draws, same boxes, same example order.  A slot too small for a
    crop makes the worker finish that picture itself (method 4, prepared)."""
    from multibox_amd.input_workers import ParallelTrainInput
    path = str(tmp_path / "t.tfrecords")
    specs = [(200 + 17 * i, 310 - 9 * i, [[.1, .2, .5, .6], [.3, .3, .9, .8]] if i % 3 else []) for i in range(8)]
    _make_records(path, specs)
    cfg = Cfg(dict(INPUT_SIZE=64, DO_RANDOM_FLIP_LEFT_RIGHT=True, DO_COLOR_DISTORTION=0.7, COLOR_DISTORT_FAST=False,
                   DO_RANDOM_CROP=0.6, RANDOM_CROP_MIN_OBJECT_COVERED=0.5, RANDOM_CROP_ASPECT_RATIO_RANGE=[0.7, 1.4],
                   RANDOM_CROP_AREA_RANGE=[0.3, 1.0], RANDOM_CROP_MAX_ATTEMPTS=50, RANDOM_CROP_MINIMUM_AREA=10,
                   DO_RANDOM_BBOX_SHIFT=0.5, RANDOM_BBOX_SHIFT_EXTENT=4))
    ref = list(I.train_batches([path], cfg, 4, 5, num_epochs=1, seed=11))
    assert len(ref) == 2

    class HostFinisher:                                     # stands in for augment.BatchAugmenter without a GPU
        prepared = 0

        def begin(self):
            self.images = []

        def add(self, pixels, method, flip, color):
            if method == 4:
                self.prepared += 1
                img = np.array(pixels, np.float32)
            else:
                p = I.AugmentPlan()
                p.crop, p.method, p.flip, p.color = None, method, flip, color
                img = I.apply_plan(np.asarray(pixels).astype(np.float32) * np.float32(1.0 / 255.0), p, 64)
            self.images.append((img - np.float32(0.5)) * np.float32(2.0))

    for max_px, want_prepared in ((1024 * 1024, False), (150 * 150, True)):
        src = ParallelTrainInput([path], cfg, 4, 5, num_workers=1, num_epochs=1, seed=11, shuffle=False,
                                 tmpdir=str(tmp_path), device_augment=True, max_source_pixels=max_px)
        fin = HostFinisher()
        for ra, rb, rn, rids in ref:
            bb, nn, ids = src.next_into(fin)
            assert ids == rids and np.array_equal(bb, rb) and np.array_equal(nn, rn)
            assert np.array_equal(np.stack(fin.images), ra)
        with pytest.raises(StopIteration):
            src.next_into(fin)
        with pytest.raises(TypeError):
            next(src)
        src.close()
        assert (fin.prepared > 0) == want_prepared


def test_detect_patch_geometry_matches_extract_patches(tmp_path):
    """inputs.patch_windows is the geometry of detect.extract_patches (detect.py:20-72, pinned to the reference's goldens);
    detect_batches(device_patches=True) carries the same metadata as the host batches, with windows instead of pixels."""
    from multibox_amd.detect import extract_patches
    rng = np.random.RandomState(0)
    for (H, W, ph, pw, sh, sw) in [(480, 640, 299, 299, 113, 113), (300, 300, 299, 299, 50, 50), (412, 500, 200, 350, 71, 150),
                                   (100, 100, 200, 200, 10, 10)]:
        img = rng.rand(H, W, 3).astype(np.float32)
        patches, offs, res, n = extract_patches(img, (ph, pw), (sh, sw))
        o2, r2 = I.patch_windows(H, W, (ph, pw), (sh, sw))
        assert len(o2) == int(n) and np.array_equal(np.array(o2, np.int32).reshape(-1, 2), offs) and np.array_equal(r2, res)
        for (y, x), p in zip(o2, patches):
            assert np.array_equal(img[y:y + ph, x:x + pw], p)
    path = str(tmp_path / "d.tfrecords")
    _make_records(path, [(320, 420, []), (300, 300, []), (412, 412, [])])
    cfg = Cfg(dict(INPUT_SIZE=299, DETECTION=dict(
        USE_ORIGINAL_IMAGE=True, ORIGINAL_IMAGE_MAX_TO_KEEP=200, USE_FLIPPED_ORIGINAL_IMAGE=True, FLIPPED_IMAGE_MAX_TO_KEEP=100,
        CROPS=[dict(HEIGHT=299, WIDTH=299, HEIGHT_STRIDE=113, WIDTH_STRIDE=113, FLIP=False, MAX_TO_KEEP=50),
               dict(HEIGHT=250, WIDTH=280, HEIGHT_STRIDE=60, WIDTH_STRIDE=90, FLIP=True, MAX_TO_KEEP=40)])))
    host = list(I.detect_batches([path], cfg, 5, keep_partial=True))
    dev = list(I.detect_batches([path], cfg, 5, keep_partial=True, device_patches=True))
    assert len(host) == len(dev) > 2
    for hb, db in zip(host, dev):
        for k in ("offsets", "dims", "is_flipped", "restrictions", "max_to_keep", "image_hw"):
            assert np.array_equal(hb[k], db[k])
        assert hb["image_ids"] == db["image_ids"] and "images" not in db and len(db["patches"]) == 5
        # finishing the windows on the host reproduces the host batch
        for i, p in enumerate(db["patches"]):
            if p is None:
                assert not hb["images"][i].any() and int(db["max_to_keep"][i, 0]) == 0
                continue
            si, (y, x, h, w), fs = p
            src = (db["sources"][si].astype(np.float32) * np.float32(1 / 255.) - np.float32(0.5)) * np.float32(2.0)
            src = src[:, ::-1] if fs else src
            assert np.array_equal(I.resize_bilinear_tf(src[y:y + h, x:x + w], 299, 299), hb["images"][i])
    assert any(p is None for p in dev[-1]["patches"])


def test_decode_process_pool_matches_thread_path(tmp_path):
    """NUM_DECODE_PROCESSES > 0 (round 4): JPEGs decoded by worker processes through one shared-memory block give the batches
    of the thread path, sources included; pictures outlive the pool (the mapping goes with the last picture), slots are
    recycled, a picture larger than a slot or a full ring falls back to decoding in this process."""
    path = str(tmp_path / "p.tfrecords")
    _make_records(path, [(320, 420, []), (300, 300, []), (412, 412, []), (200, 640, []), (480, 640, [])])
    cfg = Cfg(dict(INPUT_SIZE=299, DETECTION=dict(
        USE_ORIGINAL_IMAGE=True, ORIGINAL_IMAGE_MAX_TO_KEEP=200, USE_FLIPPED_ORIGINAL_IMAGE=True, FLIPPED_IMAGE_MAX_TO_KEEP=100,
        CROPS=[dict(HEIGHT=299, WIDTH=299, HEIGHT_STRIDE=113, WIDTH_STRIDE=113, FLIP=False, MAX_TO_KEEP=50)])))
    a = list(I.detect_batches([path], cfg, 4, keep_partial=True, device_patches=True, decode_processes=0))
    b = list(I.detect_batches([path], cfg, 4, keep_partial=True, device_patches=True, decode_processes=2))
    assert len(a) == len(b) > 2
    for x, y in zip(a, b):
        assert x["patches"] == y["patches"] and len(x["sources"]) == len(y["sources"])
        assert all(np.array_equal(s, t) for s, t in zip(x["sources"], y["sources"]))
    # the pool on its own: three slots of 256 KB for six pictures, the last of them larger than a slot
    import io
    from PIL import Image
    rng = np.random.RandomState(0)
    shapes = [(120, 160), (121, 160), (122, 160), (123, 160), (124, 160), (400, 400)]
    jpegs = []
    for h, w in shapes:
        buf = io.BytesIO()
        Image.fromarray(rng.randint(0, 255, (h, w, 3)).astype(np.uint8)).save(buf, format="JPEG", quality=95)
        jpegs.append(buf.getvalue())
    pool = I._DecodePool(2, slots=3, slot_bytes=1 << 18)
    outs = [pool.result(h) for h in [pool.submit(j) for j in jpegs]]
    assert all(np.array_equal(o, I.decode_image_u8(j)) for o, j in zip(outs, jpegs))
    pool.close()                                          # pictures stay readable after close()
    assert all(np.array_equal(o, I.decode_image_u8(j)) for o, j in zip(outs, jpegs))
    del outs
    assert pool.free.qsize() == 3
