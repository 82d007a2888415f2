"""Host-side input pipelines (SURVEY 8f, rows F1 and F3 -- first cut, no TensorFlow):

  * detect_batches(): the multi-crop patch generator of detect.input_nodes (detect.py:134-292) --
    original / flipped image resized to INPUT_SIZE, sliding crops (extract_patches, detect.py:20-72),
    per-patch metadata, batches of BATCH_SIZE patches across images; like tf.train.batch with
    enqueue_many the incomplete last batch is NOT emitted (detect.py:283-289).
  * train_batches(): inputs.input_nodes (inputs.py:200-373) WITHOUT the random augmentations (bbox shift,
    distorted crop, random resize method, colour distortion): decode, bilinear resize, optional seeded
    left-right flip, gt padding to MAX_NUM_BBOXES (inputs.py:340-351).  The augmentations are not built yet.

JPEG decoding uses PIL (libjpeg), TF used its own libjpeg build: "parity unpinned" at the bit level.
The bilinear resize restates TF-0.11's legacy kernel (align_corners=False: src = dst * in/out, no half-pixel
offset), which torch's interpolate does not provide.
"""
from __future__ import annotations

import io

import numpy as np

from . import tfrecord
from .detect import extract_patches


def decode_image(jpeg_bytes):
    """tf.image.decode_jpeg(channels=3) + convert_image_dtype(float32): [H,W,3] in [0,1]."""
    from PIL import Image
    img = Image.open(io.BytesIO(jpeg_bytes)).convert("RGB")
    return np.asarray(img, dtype=np.uint8).astype(np.float32) * np.float32(1.0 / 255.0)


def resize_bilinear_tf(img, out_h, out_w):
    """tf.image.resize_bilinear(align_corners=False) of TF 0.11 on [..., H, W, C] float32."""
    img = np.asarray(img, np.float32)
    H, W = img.shape[-3], img.shape[-2]
    ys = np.arange(out_h, dtype=np.float32) * np.float32(H / float(out_h))
    xs = np.arange(out_w, dtype=np.float32) * np.float32(W / float(out_w))
    y0 = np.floor(ys).astype(np.int64); y1 = np.minimum(y0 + 1, H - 1); yl = (ys - y0).astype(np.float32)
    x0 = np.floor(xs).astype(np.int64); x1 = np.minimum(x0 + 1, W - 1); xl = (xs - x0).astype(np.float32)
    top_l, top_r = img[..., y0, :, :][..., :, x0, :], img[..., y0, :, :][..., :, x1, :]
    bot_l, bot_r = img[..., y1, :, :][..., :, x0, :], img[..., y1, :, :][..., :, x1, :]
    xl_ = xl[:, None]
    top = top_l + (top_r - top_l) * xl_
    bot = bot_l + (bot_r - bot_l) * xl_
    return (top + (bot - top) * yl[:, None, None]).astype(np.float32)


def _records(tfrecords):
    for path in tfrecords:
        for payload in tfrecord.read_records(path):
            yield tfrecord.parse_example(payload)


def detect_patches_for_image(image01, image_hw, cfg):
    """All patches + metadata of ONE image (detect.py:183-281).  image01: decoded [H,W,3] in [0,1]."""
    S = int(cfg.INPUT_SIZE)
    det = cfg.DETECTION
    image = (image01 - np.float32(0.5)) * np.float32(2.0)            # detect.py:181-182
    flipped = image[:, ::-1]
    patches, offs, dims, flips, rests, keeps = [], [], [], [], [], []

    def add(p, o, d, f, r, k):
        patches.append(p); offs.append(o); dims.append(d); flips.append(f); rests.append(r); keeps.append(k)
    if det.get("USE_ORIGINAL_IMAGE", False):
        add(resize_bilinear_tf(image, S, S), (0, 0), tuple(image_hw), 0, (0., 0., 1., 1.), int(det.ORIGINAL_IMAGE_MAX_TO_KEEP))
    if det.get("USE_FLIPPED_ORIGINAL_IMAGE", False):
        add(resize_bilinear_tf(flipped, S, S), (0, 0), tuple(image_hw), 1, (0., 0., 1., 1.), int(det.FLIPPED_IMAGE_MAX_TO_KEEP))
    for crop in det.get("CROPS", None) or []:
        src = flipped if crop.FLIP else image
        cp, co, cr, n = extract_patches(src, (crop.HEIGHT, crop.WIDTH), (crop.HEIGHT_STRIDE, crop.WIDTH_STRIDE))
        for i in range(int(n)):
            add(resize_bilinear_tf(cp[i], S, S), tuple(co[i]), (crop.HEIGHT, crop.WIDTH), 1 if crop.FLIP else 0,
                tuple(cr[i]), int(crop.MAX_TO_KEEP))
    return patches, offs, dims, flips, rests, keeps


def detect_batches(tfrecords, cfg, batch_size, keep_partial=False):
    """Yield dicts of numpy arrays: images [B,S,S,3], offsets [B,2], dims [B,2], is_flipped [B,1],
    restrictions [B,4], max_to_keep [B,1], image_hw [B,2], image_ids [B] -- the fetches of detect.py:398-406."""
    buf = {k: [] for k in ("images", "offsets", "dims", "is_flipped", "restrictions", "max_to_keep", "image_hw", "image_ids")}

    def emit():
        out = dict(images=np.stack(buf["images"][:batch_size]).astype(np.float32),
                   offsets=np.array(buf["offsets"][:batch_size], np.int32), dims=np.array(buf["dims"][:batch_size], np.int32),
                   is_flipped=np.array(buf["is_flipped"][:batch_size], np.int32).reshape(-1, 1),
                   restrictions=np.array(buf["restrictions"][:batch_size], np.float32),
                   max_to_keep=np.array(buf["max_to_keep"][:batch_size], np.int32).reshape(-1, 1),
                   image_hw=np.array(buf["image_hw"][:batch_size], np.int32), image_ids=list(buf["image_ids"][:batch_size]))
        for k in buf:
            del buf[k][:batch_size]
        return out
    for ex in _records(tfrecords):
        img = decode_image(ex["image/encoded"][0])
        hw = (int(ex["image/height"][0]), int(ex["image/width"][0]))
        image_id = ex["image/id"][0].decode("utf-8")
        p, o, d, f, r, k = detect_patches_for_image(img, hw, cfg)
        buf["images"] += p; buf["offsets"] += o; buf["dims"] += d; buf["is_flipped"] += f
        buf["restrictions"] += r; buf["max_to_keep"] += k
        buf["image_hw"] += [hw] * len(p); buf["image_ids"] += [image_id] * len(p)
        while len(buf["images"]) >= batch_size:
            yield emit()
    if keep_partial and buf["images"]:
        n = len(buf["images"])
        pad = batch_size - n
        S = int(cfg.INPUT_SIZE)
        buf["images"] += [np.zeros((S, S, 3), np.float32)] * pad
        buf["offsets"] += [(0, 0)] * pad; buf["dims"] += [(S, S)] * pad; buf["is_flipped"] += [0] * pad
        buf["restrictions"] += [(0., 0., 1., 1.)] * pad; buf["max_to_keep"] += [0] * pad      # keep nothing of the padding
        buf["image_hw"] += [(S, S)] * pad; buf["image_ids"] += [buf["image_ids"][-1]] * pad
        yield emit()


def train_batches(tfrecords, cfg, batch_size, max_num_bboxes, num_epochs=None, seed=0):
    """Yield (images [B,S,S,3] in [-1,1], bboxes [B,G,4] x1,y1,x2,y2, num_bboxes [B] int32, image_ids)."""
    for key in ("DO_RANDOM_BBOX_SHIFT", "DO_RANDOM_CROP", "DO_COLOR_DISTORTION"):
        if float(cfg.get(key, 0) or 0) > 0:
            raise NotImplementedError("%s > 0: this augmentation of inputs.py is not built yet (SURVEY 8f F1)" % key)
    rng = np.random.RandomState(seed)
    S = int(cfg.INPUT_SIZE)
    imgs, boxes, nums, ids = [], [], [], []
    epoch = 0
    while num_epochs is None or epoch < num_epochs:
        got = False
        for ex in _records(tfrecords):
            got = True
            img = resize_bilinear_tf(decode_image(ex["image/encoded"][0]), S, S)
            n = int(ex["image/object/bbox/count"][0])
            xmin, ymin = np.array(ex.get("image/object/bbox/xmin", []), np.float32), np.array(ex.get("image/object/bbox/ymin", []), np.float32)
            xmax, ymax = np.array(ex.get("image/object/bbox/xmax", []), np.float32), np.array(ex.get("image/object/bbox/ymax", []), np.float32)
            if cfg.get("DO_RANDOM_FLIP_LEFT_RIGHT", False) and rng.uniform() < 0.5:          # inputs.py:319-323
                img = img[:, ::-1]
                xmin, xmax = np.float32(1.0) - xmax, np.float32(1.0) - xmin
            bb = np.zeros((max_num_bboxes, 4), np.float32)                                    # inputs.py:340-348
            n = min(n, max_num_bboxes)
            if n > 0:
                bb[:n] = np.stack([xmin, ymin, xmax, ymax], 1)[:n]
            imgs.append((img - np.float32(0.5)) * np.float32(2.0))                            # inputs.py:350-351
            boxes.append(bb); nums.append(n); ids.append(ex["image/id"][0].decode("utf-8"))
            if len(imgs) == batch_size:
                yield np.stack(imgs).astype(np.float32), np.stack(boxes), np.array(nums, np.int32), ids
                imgs, boxes, nums, ids = [], [], [], []
        if not got:
            return
        epoch += 1
