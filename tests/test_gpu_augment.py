"""mbx_augment_batch (row F1 on the GPU).
(1) Against the ORACLE, oracle/ref_inputs.py -- the independent per-pixel restatement of the reference's augmentation ops
    (crop, tf.image.resize_images with the drawn method, distort_color, flip, scaling; inputs.py:44-98, 272-351): bilinear /
    nearest bit-exact (TF's float32 kernel order is the definition), bicubic / area / colour within 2e-6 on [-1, 1]
    (float64 on both sides, different summation orders) -- test_device_path_against_the_oracle.
(2) Against the product's own HOST path (multibox_amd/inputs.py apply_plan), which the workers run when
    INPUT_AUGMENT_ON_DEVICE is false: the two product paths must agree bit for bit on resize-only items and within
    1e-6 with colour ops, at full picture sizes the per-pixel oracle is too slow for."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _plan(method, color, flip, crop=None):
    from multibox_amd import inputs as I
    p = I.AugmentPlan()
    p.crop, p.method, p.color, p.flip = crop, method, color, flip
    return p


def _host(u8, plan, S):
    from multibox_amd import inputs as I
    img = I.apply_plan(u8.astype(np.float32) * np.float32(1.0 / 255.0), plan, S)
    return (img - np.float32(0.5)) * np.float32(2.0)


def _run(cases, S):
    import torch
    from multibox_amd import inputs as I
    from multibox_amd.augment import BatchAugmenter
    aug = BatchAugmenter(len(cases), S, slot_bytes=1 << 21)
    aug.begin()
    for u8, plan in cases:
        aug.add(I.crop_pixels(u8, plan), plan.method, plan.flip, plan.color)
    out = aug.run()
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_device_path_against_the_oracle():
    from multibox_amd import inputs as I
    from oracle import ref_inputs as O
    rng = np.random.RandomState(21)
    S = 48
    cases = []
    for (h, w) in [(97, 61), (48, 48), (30, 130), (75, 40)]:
        u8 = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        for method in range(4):
            cases.append((u8, _plan(method, [], bool(rng.randint(2)))))
    u8 = rng.randint(0, 256, (90, 120, 3)).astype(np.uint8)
    for method in range(4):                                   # crops: the worker hands over only the window
        cases.append((u8, _plan(method, [], False, crop=(11, 17, 61, 83))))
    for ordering in range(4):                                 # the four colour orders of inputs.py:71-91, then the fast ones
        u8 = rng.randint(0, 256, (rng.randint(40, 90), rng.randint(40, 90), 3)).astype(np.uint8)
        cases.append((u8, _plan(ordering, I.color_ops(ordering, False, rng), bool(ordering & 1))))
    for ordering in range(2):
        u8 = rng.randint(0, 256, (50, 70, 3)).astype(np.uint8)
        cases.append((u8, _plan(0, I.color_ops(ordering, True, rng), False)))
    got = _run(cases, S)
    for i, (u8, plan) in enumerate(cases):
        want = O.augment_pixels(u8.astype(np.float32) * np.float32(1.0 / 255.0), plan.crop, plan.method, plan.color, plan.flip,
                                S, scale_to_pm1=True)
        if plan.method in (O.BILINEAR, O.NEAREST) and not plan.color:
            assert np.array_equal(got[i], want), "case %d method %d: max diff %g" % (i, plan.method, np.abs(got[i] - want).max())
        else:
            np.testing.assert_allclose(got[i], want, rtol=0, atol=2e-6, err_msg="case %d" % i)


@pytest.mark.parametrize("S", [299, 64])
def test_resize_methods_bit_identical_to_the_host_arithmetic(S):
    rng = np.random.RandomState(3)
    cases = []
    for (h, w) in [(480, 640), (333, 500), (97, 61), (299, 299), (1000, 37), (20, 20), (640, 427)]:
        u8 = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        for method in range(4):
            cases.append((u8, _plan(method, [], bool(rng.randint(2)))))
    # crops: the worker hands over only the window
    u8 = rng.randint(0, 256, (375, 500, 3)).astype(np.uint8)
    for method in range(4):
        cases.append((u8, _plan(method, [], False, crop=(31, 57, 201, 333))))
    got = _run(cases, S)
    for i, (u8, plan) in enumerate(cases):
        want = _host(u8, plan, S)
        assert got[i].shape == want.shape
        if plan.method == 2:                                    # float64 taps: the device's fmul/fadd order is the host's,
            np.testing.assert_allclose(got[i], want, rtol=0, atol=2.4e-7, err_msg="case %d" % i)   # 1 ulp at 2.0
        else:
            assert np.array_equal(got[i], want), "case %d method %d: max diff %g" % (
                i, plan.method, np.abs(got[i] - want).max())


def test_colour_orders_and_flip_match_the_host_arithmetic():
    from multibox_amd import inputs as I
    rng = np.random.RandomState(5)
    S = 299
    cases = []
    for ordering in range(4):
        for fast in (False, True):
            u8 = rng.randint(0, 256, (rng.randint(100, 500), rng.randint(100, 500), 3)).astype(np.uint8)
            if ordering == 3 and fast:
                u8[:] = u8[:1, :1]                              # a flat picture: hue undefined (d = 0), saturation 0
            color = I.color_ops(ordering, fast, rng)
            cases.append((u8, _plan(int(rng.randint(4)), color, bool(rng.randint(2)))))
    # extreme arguments: saturated brightness, hue wrap, contrast both ways
    u8 = rng.randint(0, 256, (240, 320, 3)).astype(np.uint8)
    cases.append((u8, _plan(0, [(I.COLOR_BRIGHTNESS, 32 / 255.), (I.COLOR_HUE, 0.2), (I.COLOR_CONTRAST, 1.5),
                                (I.COLOR_SATURATION, 1.5)], True)))
    cases.append((u8, _plan(2, [(I.COLOR_CONTRAST, 0.5), (I.COLOR_HUE, -0.2), (I.COLOR_BRIGHTNESS, -32 / 255.),
                                (I.COLOR_SATURATION, 0.5)], False)))
    got = _run(cases, S)
    for i, (u8, plan) in enumerate(cases):
        want = _host(u8, plan, S)
        np.testing.assert_allclose(got[i], want, rtol=0, atol=1e-6, err_msg="case %d" % i)
        assert got[i].min() >= -1.0 and got[i].max() <= 1.0     # distort_color clips to [0,1] (inputs.py:96-98)


def test_prepared_pictures_pass_through_and_partial_batches():
    import torch
    from multibox_amd.augment import BatchAugmenter, METHOD_PREPARED
    rng = np.random.RandomState(7)
    S = 32
    aug = BatchAugmenter(4, S, slot_bytes=4096)
    aug.begin()
    f = rng.rand(S, S, 3).astype(np.float32)
    aug.add(f, METHOD_PREPARED, False, [])
    u8 = rng.randint(0, 256, (S, S, 3)).astype(np.uint8)
    aug.add(u8, 1, True, [])
    out = aug.run()
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    assert out.shape == (2, S, S, 3)
    assert np.array_equal(out[0], (f - np.float32(0.5)) * np.float32(2.0))
    assert np.array_equal(out[1], ((u8.astype(np.float32) * np.float32(1 / 255.))[:, ::-1] - np.float32(0.5)) * np.float32(2.0))
    with pytest.raises(ValueError):
        aug.add(np.zeros((S, S), np.uint8), 0, False, [])
    with pytest.raises(ValueError):
        aug.add(u8, 7, False, [])


def test_augment_batch_rejects_bad_arguments():
    from multibox_amd import _lib
    L = _lib.lib()
    assert L.mbx_augment_batch(None, None, 2, 299, 0, None, None, None) == -1
    assert L.mbx_augment_batch(None, None, 0, 299, 0, None, None, None) == 0
    assert L.mbx_augment_workspace_bytes(64, 299) >= 64 * 299 * 299 * 3 * 4


def test_worker_ring_to_device_batches_match_the_host_pipeline(tmp_path):
    """ParallelTrainInput(device_augment=True) -> DevicePrefetcher (side-stream upload + mbx_augment_batch): the batches
    on the device equal inputs.train_batches() of the same seed -- boxes and ids exactly, pixels within 1e-6."""
    import torch
    from multibox_amd import inputs as I
    from multibox_amd.config import Cfg
    from multibox_amd.input_workers import ParallelTrainInput, DevicePrefetcher
    from tests.test_inputs_cpu import _make_records
    path = str(tmp_path / "t.tfrecords")
    _make_records(path, [(220 + 23 * i, 400 - 13 * i, [[.1, .2, .5, .6], [.3, .3, .9, .8]][: i % 3]) for i in range(12)])
    cfg = Cfg(dict(INPUT_SIZE=299, DO_RANDOM_FLIP_LEFT_RIGHT=True, DO_COLOR_DISTORTION=0.7, COLOR_DISTORT_FAST=False,
                   DO_RANDOM_CROP=0.6, RANDOM_CROP_MIN_OBJECT_COVERED=0.5, RANDOM_CROP_ASPECT_RATIO_RANGE=[0.7, 1.4],
                   RANDOM_CROP_AREA_RANGE=[0.3, 1.0], RANDOM_CROP_MAX_ATTEMPTS=50, RANDOM_CROP_MINIMUM_AREA=10,
                   DO_RANDOM_BBOX_SHIFT=0.5, RANDOM_BBOX_SHIFT_EXTENT=4))
    ref = list(I.train_batches([path], cfg, 4, 5, num_epochs=1, seed=21))
    src = ParallelTrainInput([path], cfg, 4, 5, num_workers=1, num_epochs=1, seed=21, shuffle=False,
                             tmpdir=str(tmp_path), device_augment=True)
    pre = DevicePrefetcher(src, 4, 299, 5, device="cuda", depth=2)
    got = []
    while True:
        try:
            images, bb, n, ids = pre.next()
        except StopIteration:
            break
        torch.cuda.current_stream().synchronize()
        got.append((images.cpu().numpy().copy(), bb.cpu().numpy().copy(), n.cpu().numpy().copy(), list(ids)))
    pre.close()
    assert len(got) == len(ref) == 3
    for (a, b, n, ids), (ra, rb, rn, rids) in zip(got, ref):
        assert ids == rids and np.array_equal(b, rb) and np.array_equal(n, rn)
        np.testing.assert_allclose(a, ra, rtol=0, atol=1e-6)


def test_detect_patches_on_the_device_are_bit_identical_to_the_host_input(tmp_path):
    """mbx_extract_patches behind augment.PatchExtractor against inputs.detect_batches (detect.py:181-281: scale to
    [-1,1], mirror, sliding windows, legacy bilinear resize): bit-identical pictures, padding entries all zero."""
    import torch
    from multibox_amd import inputs as I
    from multibox_amd.augment import PatchExtractor
    from multibox_amd.config import Cfg
    from tests.test_inputs_cpu import _make_records
    path = str(tmp_path / "d.tfrecords")
    _make_records(path, [(320, 420, []), (300, 300, []), (412, 500, []), (640, 480, [])])
    cfg = Cfg(dict(INPUT_SIZE=299, DETECTION=dict(
        USE_ORIGINAL_IMAGE=True, ORIGINAL_IMAGE_MAX_TO_KEEP=200, USE_FLIPPED_ORIGINAL_IMAGE=True, FLIPPED_IMAGE_MAX_TO_KEEP=100,
        CROPS=[dict(HEIGHT=299, WIDTH=299, HEIGHT_STRIDE=113, WIDTH_STRIDE=113, FLIP=False, MAX_TO_KEEP=50),
               dict(HEIGHT=250, WIDTH=280, HEIGHT_STRIDE=60, WIDTH_STRIDE=90, FLIP=True, MAX_TO_KEEP=40)])))
    B = 8
    host = list(I.detect_batches([path], cfg, B, keep_partial=True))
    dev = list(I.detect_batches([path], cfg, B, keep_partial=True, device_patches=True))
    ex = PatchExtractor(B, 299, capacity_bytes=1 << 20)         # small: the first big batch makes it grow
    assert len(host) == len(dev) >= 4
    for hb, db in zip(host, dev):
        got = ex(db["sources"], db["patches"])
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), hb["images"])
    with pytest.raises(ValueError):
        ex(db["sources"], [(0, (0, 0, 10 ** 6, 10), 0)] * B)


def test_eval_images_on_the_device_are_bit_identical_to_the_host_input(tmp_path):
    """eval_inputs.py:20-115 (decode, legacy bilinear resize, [-1,1]) through BatchAugmenter method 0."""
    import torch
    from multibox_amd import inputs as I
    from multibox_amd.augment import BatchAugmenter
    from multibox_amd.config import Cfg
    from tests.test_inputs_cpu import _make_records
    path = str(tmp_path / "e.tfrecords")
    _make_records(path, [(320 + 30 * i, 420 - 25 * i, [[.1, .2, .5, .6]]) for i in range(6)])
    cfg = Cfg(dict(INPUT_SIZE=299))
    host = list(I.eval_batches([path], cfg, 3, 5))
    dev = list(I.eval_batches([path], cfg, 3, 5, device_images=True))
    aug = BatchAugmenter(3, 299, slot_bytes=1 << 16)            # too small on purpose: add() grows the staging
    assert len(host) == len(dev) == 2
    for (hi, hb, hn, ha, hids), (di, db, dn, da, dids) in zip(host, dev):
        aug.begin()
        for u8 in di:
            aug.add(u8, 0, False, [])
        got = aug.run()
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), hi)
        assert np.array_equal(hb, db) and np.array_equal(hn, dn) and np.array_equal(ha, da) and hids == dids
