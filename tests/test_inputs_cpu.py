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
            "image/object/bbox/count": [len(boxes)]}))
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
    with pytest.raises(NotImplementedError):
        next(I.train_batches([path], Cfg(dict(INPUT_SIZE=299, DO_RANDOM_CROP=0.5)), 2, 5))
