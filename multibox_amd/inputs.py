"""Host-side input pipelines (no TensorFlow):

  * detect_batches(): the multi-crop patch generator of detect.input_nodes (detect.py:134-292) --
    original / flipped image resized to INPUT_SIZE, sliding crops (extract_patches, detect.py:20-72),
    per-patch metadata, batches of BATCH_SIZE patches across images; like tf.train.batch with
    enqueue_many the incomplete last batch is NOT emitted (detect.py:283-289).
  * train_batches(): inputs.input_nodes (inputs.py:200-373): decode, random bbox shift
    (inputs.py:184-203), distorted crop around the boxes (inputs.py:100-182 over the algorithm of TF's
    sample_distorted_bounding_box op), resize by a randomly chosen method (bilinear / nearest / bicubic /
    area, inputs.py:296-302), colour distortion (inputs.py:44-98, 312-320), left-right flip
    (inputs.py:323-327), gt padding to MAX_NUM_BBOXES and [-1,1] scaling (inputs.py:338-351), shuffle
    buffer (capacity / min_after_dequeue, inputs.py:353-362).

Parity: the deterministic arithmetic (box shift clipping, crop -> box transform with the reference's clip-to-
IMAGE-size quirk and minimum-area filter, padding, scaling) is restated exactly and tested; the RANDOM draws
come from numpy's RandomState, not TF's Philox streams, and JPEG decoding uses PIL (libjpeg) where TF used
its own build: "parity unpinned" at the bit level.  The bilinear resize restates TF-0.11's legacy kernel
(align_corners=False: src = dst * in/out, no half-pixel offset), which torch's interpolate does not provide;
nearest / bicubic / area follow the same legacy coordinate rule.
"""
from __future__ import annotations

import io

import numpy as np

from . import tfrecord


def extract_patches(*args, **kwargs):
    """multibox_amd.detect.extract_patches (detect.py:20-72), imported on first use: that module imports torch, which the
    training input worker processes (input_workers.py) neither need nor should load."""
    from .detect import extract_patches as f
    return f(*args, **kwargs)


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


def patch_windows(H, W, patch_dims, strides, non_edge_restriction=0.1):
    """The geometry of detect.extract_patches (detect.py:20-72) without the pixels: [(y, x)...] offsets and the
    restriction rows, in its order (tests/test_inputs_cpu.py holds it against extract_patches)."""
    ph, pw = int(patch_dims[0]), int(patch_dims[1])
    offs, res = [], []
    for h in range(0, H - ph + 1, int(strides[0])):
        for w in range(0, W - pw + 1, int(strides[1])):
            offs.append((h, w))
            res.append((0.0 if w == 0 else non_edge_restriction, 0.0 if h == 0 else non_edge_restriction,
                        1.0 if w + pw == W else 1.0 - non_edge_restriction,
                        1.0 if h + ph == H else 1.0 - non_edge_restriction))
    return offs, np.array(res, np.float32).reshape(-1, 4)


def detect_patch_plan(H, W, image_hw, cfg):
    """Every patch of ONE decoded H x W image (detect.py:183-281) as geometry: a list of
    (window (y, x, h, w), flip_source, offset (y, x), dims (h, w), is_flipped, restrictions, max_to_keep); the pixels of
    patch i are resize_bilinear(((image - 0.5) * 2)[:, ::-1 if flip_source][window], S, S)."""
    det = cfg.DETECTION
    plan = []
    if det.get("USE_ORIGINAL_IMAGE", False):
        plan.append(((0, 0, H, W), 0, (0, 0), tuple(image_hw), 0, (0., 0., 1., 1.), int(det.ORIGINAL_IMAGE_MAX_TO_KEEP)))
    if det.get("USE_FLIPPED_ORIGINAL_IMAGE", False):
        plan.append(((0, 0, H, W), 1, (0, 0), tuple(image_hw), 1, (0., 0., 1., 1.), int(det.FLIPPED_IMAGE_MAX_TO_KEEP)))
    for crop in det.get("CROPS", None) or []:
        offs, res = patch_windows(H, W, (crop.HEIGHT, crop.WIDTH), (crop.HEIGHT_STRIDE, crop.WIDTH_STRIDE))
        f = 1 if crop.FLIP else 0
        for (y, x), r in zip(offs, res):
            plan.append(((y, x, int(crop.HEIGHT), int(crop.WIDTH)), f, (y, x), (int(crop.HEIGHT), int(crop.WIDTH)), f,
                         tuple(float(v) for v in r), int(crop.MAX_TO_KEEP)))
    return plan


def detect_patches_for_image(image01, image_hw, cfg):
    """All patches + metadata of ONE image (detect.py:183-281) on the host.  image01: decoded [H,W,3] in [0,1]."""
    S = int(cfg.INPUT_SIZE)
    image = (image01 - np.float32(0.5)) * np.float32(2.0)            # detect.py:181-182
    flipped = image[:, ::-1]
    patches, offs, dims, flips, rests, keeps = [], [], [], [], [], []
    for (y, x, h, w), fs, o, d, f, r, k in detect_patch_plan(image.shape[0], image.shape[1], image_hw, cfg):
        patches.append(resize_bilinear_tf((flipped if fs else image)[y:y + h, x:x + w], S, S))
        offs.append(o); dims.append(d); flips.append(f); rests.append(r); keeps.append(k)
    return patches, offs, dims, flips, rests, keeps


# ---- JPEG decoding in worker PROCESSES (round 4): the decoded pixels come back through one shared-memory block
_SHM = None                       # worker side: the attached block


def _decode_worker_init(shm_name):
    global _SHM
    from multiprocessing import shared_memory
    _SHM = shared_memory.SharedMemory(name=shm_name)


def _decode_into_shm(offset, capacity, jpeg_bytes):
    """Worker: decode one JPEG into the shared block at `offset`; (H, W), or None if it needs more than `capacity` bytes."""
    u8 = decode_image_u8(jpeg_bytes)
    if u8.size > capacity:
        return None
    np.frombuffer(_SHM.buf, np.uint8, u8.size, offset)[:] = u8.reshape(-1)
    return int(u8.shape[0]), int(u8.shape[1])


class _DecodePool:
    """`processes` spawned workers (no torch, no GPU) decode JPEGs into slots of one shared-memory block; the parent wraps a
    finished slot as a numpy view whose finalizer gives the slot back, so a picture lives exactly as long as something
    (a pending patch, a batch being staged) refers to it.  No free slot, or a picture larger than a slot: decoded in
    this process instead (never a wait: the consumer of the slots is the caller itself)."""

    def __init__(self, processes, slots=160, slot_bytes=3 << 19):
        import multiprocessing as mp
        import queue
        import shutil
        from concurrent.futures import ProcessPoolExecutor
        from multiprocessing import shared_memory
        # the block lives in /dev/shm: size it from what is free there (the 64 MB default of many containers holds 40 slots,
        # not 160 -- a worker writing past it would die of SIGBUS, not fall back); fewer slots only mean more pictures
        # decoded in this process
        # ... and every rank of the node (and every pool of a rank) sees the SAME free figure when they start together: the share
        # is divided by the local world size, and the pages are RESERVED up front (posix_fallocate: tmpfs allocates lazily, so
        # a block that merely fits at creation can still end in SIGBUS on a worker's first write -- ADVICE round 5); a block
        # that cannot be reserved is halved until it can
        import os as _os
        sharers = max(1, int(_os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
        try:
            avail = shutil.disk_usage("/dev/shm").free
            slots = min(int(slots), max(1, int(0.5 * avail / sharers) // int(slot_bytes)))
        except OSError:
            pass
        self.slot_bytes, self._closing = int(slot_bytes), False
        self._returned = set()                                  # slots close() has handed back (a late result() must not return them twice)
        while True:
            shm = shared_memory.SharedMemory(create=True, size=int(slots) * self.slot_bytes)
            try:
                _os.posix_fallocate(shm._fd, 0, int(slots) * self.slot_bytes)
                break
            except (OSError, AttributeError):
                shm.close()
                shm.unlink()
                if slots <= 1:
                    shm = shared_memory.SharedMemory(create=True, size=self.slot_bytes)     # (one slot, lazily: the old behaviour)
                    slots = 1
                    break
                slots = max(1, int(slots) // 2)
        self.slots = int(slots)
        self.shm = shm
        self.free = queue.SimpleQueue()
        self.out = {}                                           # slot -> its decode future, until result() has taken it
        for i in range(int(slots)):
            self.free.put(i)
        self.pool = ProcessPoolExecutor(max_workers=int(processes), mp_context=mp.get_context("spawn"),
                                        initializer=_decode_worker_init, initargs=(self.shm.name,))

    def submit(self, jpeg_bytes):
        """-> a handle for result()."""
        try:
            slot = self.free.get_nowait()
        except Exception:                                       # (queue.Empty) every slot is referenced: decode here, later
            return (None, jpeg_bytes)
        try:
            fut = self.pool.submit(_decode_into_shm, slot * self.slot_bytes, self.slot_bytes, jpeg_bytes)
        except Exception:                                       # (BrokenProcessPool: a worker died) decode in this process
            self.free.put(slot)
            return (None, jpeg_bytes)
        self.out[slot] = fut
        return (slot, fut, jpeg_bytes)

    def result(self, handle):
        import weakref
        if handle[0] is None:
            return decode_image_u8(handle[1])
        slot, fut, jpeg_bytes = handle
        self.out.pop(slot, None)
        try:
            hw = fut.result()
        except Exception:                                       # a failed / cancelled decode: the slot comes back, the picture is decoded here
            hw = None
        if hw is None:                                          # larger than a slot
            if slot not in self._returned:                      # (close() may have handed it back already)
                self._release(slot)
            return decode_image_u8(jpeg_bytes)
        arr = np.ndarray((hw[0], hw[1], 3), np.uint8, buffer=self.shm.buf, offset=slot * self.slot_bytes)
        # views of `arr` keep it alive through .base; the bound method keeps THIS object -- and with it the mapping -- alive
        # as long as a picture is (numpy takes no buffer export: SharedMemory.close() would unmap under a live array)
        weakref.finalize(arr, self._release, slot)
        return arr

    def _release(self, slot):
        self.free.put(slot)
        if self._closing and self.free.qsize() >= self.slots:
            self.shm.close()

    def close(self):
        """Stop the workers and remove the block's name; the mapping itself goes when the last picture does.  Slots whose
        decode was still pending (an early exit of the consumer: their futures are cancelled and result() never comes) are
        handed back here -- without that the count never reached `slots` and the mapping leaked."""
        self.pool.shutdown(wait=True, cancel_futures=True)
        try:
            self.shm.unlink()
        except FileNotFoundError:
            pass
        self._closing = True
        for slot in list(self.out):
            self.out.pop(slot, None)
            self._returned.add(slot)
            self.free.put(slot)
        if self.free.qsize() >= self.slots:
            self.shm.close()


def _decoded_ahead(planned, threads, window=32, processes=0):
    """(example, plan, mine, decoded uint8 image or None) in record order for _planned_examples() items: the JPEGs of the
    records this rank needs are decoded ahead of the consumer -- by `threads` threads (PIL releases the GIL while it decodes)
    or, on request, by `processes` worker processes through shared memory (_DecodePool); the others pass through undecoded."""
    if processes > 0:
        from collections import deque
        pool = _DecodePool(processes)
        try:
            pending = deque()
            for ex, plan, mine in planned:
                pending.append((ex, plan, mine, pool.submit(ex["image/encoded"][0]) if mine else None))
                if len(pending) >= window:
                    e, pl, m, h = pending.popleft()
                    yield e, pl, m, (pool.result(h) if h is not None else None)
            while pending:
                e, pl, m, h = pending.popleft()
                yield e, pl, m, (pool.result(h) if h is not None else None)
        finally:
            pool.close()
        return
    if threads <= 1:
        for ex, plan, mine in planned:
            yield ex, plan, mine, (decode_image_u8(ex["image/encoded"][0]) if mine else None)
        return
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as pool:
        pending = deque()
        for ex, plan, mine in planned:
            pending.append((ex, plan, mine, pool.submit(decode_image_u8, ex["image/encoded"][0]) if mine else None))
            if len(pending) >= window:
                e, pl, m, f = pending.popleft()
                yield e, pl, m, (f.result() if f is not None else None)
        while pending:
            e, pl, m, f = pending.popleft()
            yield e, pl, m, (f.result() if f is not None else None)


def jpeg_size(jpeg_bytes):
    """(height, width) from the JPEG header alone (PIL parses the header on open; nothing is decoded)."""
    from PIL import Image
    with Image.open(io.BytesIO(jpeg_bytes)) as im:
        w, h = im.size
    return int(h), int(w)


def _planned_examples(examples, cfg, batch_size, rank, world, stats):
    """(example, patch plan, mine) in record order.  `mine`: at least one patch of the record falls into a batch this
    rank owns (batch i of the global patch stream belongs to rank i % world).  The plan only needs the picture's size,
    which comes from the JPEG header: records that are not `mine` are never decoded."""
    g = 0
    for ex in examples:
        hw = (int(ex["image/height"][0]), int(ex["image/width"][0]))
        H, W = jpeg_size(ex["image/encoded"][0])
        plan = detect_patch_plan(H, W, hw, cfg)
        n = len(plan)
        mine = world == 1 or (n > 0 and any(b % world == rank for b in range(g // batch_size, (g + n - 1) // batch_size + 1)))
        g += n
        if stats is not None:
            stats["records"] = stats.get("records", 0) + 1
            stats["decoded"] = stats.get("decoded", 0) + int(mine)
        yield ex, plan, mine


def detect_batches(tfrecords, cfg, batch_size, keep_partial=False, device_patches=False, decode_threads=None, rank=0,
                   world=1, stats=None, decode_processes=None):
    """Yield dicts of numpy arrays: images [B,S,S,3], offsets [B,2], dims [B,2], is_flipped [B,1],
    restrictions [B,4], max_to_keep [B,1], image_hw [B,2], image_ids [B] -- the fetches of detect.py:398-406 -- plus
    "batch_index", the batch's position in the single-process stream (tf.train.batch order, detect.py:283-292).
    rank / world (ranks write disjoint result lists): only the batches with batch_index % world == rank are
    yielded, and only the records that contribute a patch to one of them are DECODED -- the others are walked by their
    header (size -> number of patches) so that the batching stays exactly the single-process one.  stats (a dict)
    counts "records" seen and "decoded".
    device_patches=True leaves the pixels to the GPU (mbx_extract_patches): instead of "images" a batch carries
    "sources" (the decoded uint8 images it touches) and "patches" [(source index, window, flip_source) or None for
    padding]: multibox_amd.augment.PatchExtractor turns them into the [B,S,S,3] tensor on the device.  The JPEGs are
    then decoded ahead of the consumer, in record order: by decode_threads threads (default NUM_INPUT_THREADS) or, with
    decode_processes > 0 (NUM_DECODE_PROCESSES, default 0), by that many worker processes through shared memory -- measured
    SLOWER end to end (8192 VGA records, BATCH_SIZE 256: 4 threads 16 300 patches/s, 4 processes 15 700, 8 processes 15 500,
    8 threads 13 800): with the pixels' staging copy in the producer thread the decode is not what limits detect.py."""
    buf = {k: [] for k in ("images", "offsets", "dims", "is_flipped", "restrictions", "max_to_keep", "image_hw", "image_ids")}
    S = int(cfg.INPUT_SIZE)

    def emit():
        out = dict(offsets=np.array(buf["offsets"][:batch_size], np.int32), dims=np.array(buf["dims"][:batch_size], np.int32),
                   is_flipped=np.array(buf["is_flipped"][:batch_size], np.int32).reshape(-1, 1),
                   restrictions=np.array(buf["restrictions"][:batch_size], np.float32),
                   max_to_keep=np.array(buf["max_to_keep"][:batch_size], np.int32).reshape(-1, 1),
                   image_hw=np.array(buf["image_hw"][:batch_size], np.int32), image_ids=list(buf["image_ids"][:batch_size]))
        if device_patches:
            sources, index, patches = [], {}, []
            for item in buf["images"][:batch_size]:
                if item is None:
                    patches.append(None)
                    continue
                u8, window, fs = item
                if id(u8) not in index:
                    index[id(u8)] = len(sources)
                    sources.append(u8)
                patches.append((index[id(u8)], window, fs))
            out["sources"], out["patches"] = sources, patches
        else:
            out["images"] = np.stack(buf["images"][:batch_size]).astype(np.float32)
        for k in buf:
            del buf[k][:batch_size]
        return out
    threads = int(decode_threads if decode_threads is not None else cfg.get("NUM_INPUT_THREADS", 4)) if device_patches else 1
    procs = int(decode_processes if decode_processes is not None else cfg.get("NUM_DECODE_PROCESSES", 0)) if device_patches else 0
    planned = _planned_examples(_records(tfrecords), cfg, batch_size, rank, world, stats)
    if world == 1 and not device_patches:
        decoded = ((ex, plan, True, None) for ex, plan, _ in planned)
    else:
        decoded = _decoded_ahead(planned, threads, processes=procs)
    next_index = [0]

    def emit_mine():
        """Cut the next batch off the buffers; None if another rank owns it (its pixels were never produced)."""
        i = next_index[0]
        next_index[0] += 1
        if i % world != rank:
            for k in buf:
                del buf[k][:batch_size]
            return None
        out = emit()
        out["batch_index"] = i
        return out
    for ex, plan, mine, u8 in decoded:
        hw = (int(ex["image/height"][0]), int(ex["image/width"][0]))
        image_id = ex["image/id"][0].decode("utf-8")
        o, d, f, r, k = ([e[i] for e in plan] for i in (2, 3, 4, 5, 6))
        if not mine:
            p = [None] * len(plan)                     # placeholders: these patches only ever land in other ranks' batches
        elif device_patches:
            p = [(u8, win, fs) for win, fs, _, _, _, _, _ in plan]
        else:
            image01 = (u8.astype(np.float32) * np.float32(1.0 / 255.0)) if u8 is not None else decode_image(ex["image/encoded"][0])
            p, o, d, f, r, k = detect_patches_for_image(image01, hw, cfg)
        buf["images"] += p; buf["offsets"] += o; buf["dims"] += d; buf["is_flipped"] += f
        buf["restrictions"] += r; buf["max_to_keep"] += k
        buf["image_hw"] += [hw] * len(p); buf["image_ids"] += [image_id] * len(p)
        while len(buf["images"]) >= batch_size:
            b = emit_mine()
            if b is not None:
                yield b
    if keep_partial and buf["images"]:
        n = len(buf["images"])
        pad = batch_size - n
        buf["images"] += [None if device_patches else np.zeros((S, S, 3), np.float32)] * pad
        buf["offsets"] += [(0, 0)] * pad; buf["dims"] += [(S, S)] * pad; buf["is_flipped"] += [0] * pad
        buf["restrictions"] += [(0., 0., 1., 1.)] * pad; buf["max_to_keep"] += [0] * pad      # keep nothing of the padding
        buf["image_hw"] += [(S, S)] * pad; buf["image_ids"] += [buf["image_ids"][-1]] * pad
        b = emit_mine()
        if b is not None:
            yield b


# ------------------------------------------------------------------ other resize methods (legacy TF kernels)
def prefetched(iterable, depth=3):
    """The items of `iterable`, produced by a background thread up to `depth` items ahead of the consumer (the reference's
    input queues do the same for the detect graph, detect.py:283-292: tf.train.batch with its own runner thread).  Order is
    kept; an exception of the producer is re-raised in the consumer at the point where it happened."""
    import queue
    import threading
    q = queue.Queue(maxsize=max(int(depth), 1))
    end = object()

    def run():
        try:
            for item in iterable:
                q.put((item, None))
        except BaseException as e:       # noqa: BLE001 -- handed over, re-raised by the consumer
            q.put((None, e))
            return
        q.put((end, None))
    threading.Thread(target=run, name="mbx-input-prefetch", daemon=True).start()
    while True:
        item, err = q.get()
        if err is not None:
            raise err
        if item is end:
            return
        yield item


def resize_nearest_tf(img, out_h, out_w):
    """tf.image.resize_nearest_neighbor(align_corners=False), TF 0.11: src = min(floor(dst * in/out), in-1)."""
    img = np.asarray(img, np.float32)
    H, W = img.shape[0], img.shape[1]
    ys = np.minimum(np.floor(np.arange(out_h, dtype=np.float32) * np.float32(H / float(out_h))).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(out_w, dtype=np.float32) * np.float32(W / float(out_w))).astype(np.int64), W - 1)
    return img[ys][:, xs]


def _cubic_weights(frac, a=-0.75):
    """Keys cubic convolution coefficients (A = -0.75, TF's resize_bicubic) for taps at -1, 0, +1, +2."""
    t = frac.astype(np.float64)
    w0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    w1 = ((a + 2) * t - (a + 3)) * t * t + 1
    w2 = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    w3 = ((a * (2 - t) - 5 * a) * (2 - t) + 8 * a) * (2 - t) - 4 * a
    return np.stack([w0, w1, w2, w3], 0)


def resize_bicubic_tf(img, out_h, out_w):
    """tf.image.resize_bicubic(align_corners=False), TF 0.11: src = dst * in/out, taps clamped to the image
    (the kernel's 1024-entry coefficient table is replaced by the closed form: < 1e-3 difference)."""
    img = np.asarray(img, np.float32)

    def axis(x, n_in, n_out, ax):
        pos = np.arange(n_out, dtype=np.float32) * np.float32(n_in / float(n_out))
        i0 = np.floor(pos).astype(np.int64)
        w = _cubic_weights(pos - i0)                                     # [4, n_out]
        out = 0.0
        for k in range(4):
            idx = np.clip(i0 - 1 + k, 0, n_in - 1)
            shape = [1] * x.ndim
            shape[ax] = n_out
            out = out + np.take(x, idx, axis=ax).astype(np.float64) * w[k].reshape(shape)
        return out
    return axis(axis(img, img.shape[0], out_h, 0), img.shape[1], out_w, 1).astype(np.float32)


def resize_area_tf(img, out_h, out_w):
    """tf.image.resize_area: every output pixel is the area-weighted mean of the source rectangle
    [dst*scale, (dst+1)*scale) (fractional coverage of the border pixels).  Separable; an output pixel covers at most
    ceil(scale) + 1 source pixels, so each axis is that many weighted gathers (a dense [out, in] weight matrix product
    cost 0.2 s per image and starved the input workers)."""
    img = np.asarray(img, np.float32)

    def axis(x, n_in, n_out, ax):
        scale = n_in / float(n_out)
        o = np.arange(n_out, dtype=np.float64)
        lo, hi = o * scale, (o + 1) * scale
        i0 = np.floor(lo).astype(np.int64)
        taps = int(np.ceil(scale)) + 1
        idx = i0[None, :] + np.arange(taps)[:, None]                                   # [taps, n_out]
        w = np.minimum(hi[None, :], idx + 1.0) - np.maximum(lo[None, :], idx.astype(np.float64))
        w = np.where((idx < n_in) & (w > 0), w, 0.0)
        w = (w / np.maximum(w.sum(0, keepdims=True), 1e-12)).astype(np.float32)
        idx = np.minimum(idx, n_in - 1)
        shape = [1] * x.ndim
        shape[ax] = n_out
        out = np.take(x, idx[0], axis=ax) * w[0].reshape(shape)
        for k in range(1, taps):
            out += np.take(x, idx[k], axis=ax) * w[k].reshape(shape)
        return out
    return axis(axis(img, img.shape[0], out_h, 0), img.shape[1], out_w, 1).astype(np.float32)


RESIZE_METHODS = (resize_bilinear_tf, resize_nearest_tf, resize_bicubic_tf, resize_area_tf)   # tf.image.ResizeMethod 0..3


# ------------------------------------------------------------------------------- box augmentations
def shift_bboxes(xmin, ymin, xmax, ymax, image_height, image_width, max_num_pixels_to_shift, rng):
    """distorted_shifted_bounding_box (inputs.py:184-203): every side moves OUTWARDS by U[0, extent pixels), clipped
    to [0, 1]."""
    mw = np.float32(1.0 / float(image_width) * max_num_pixels_to_shift)
    mh = np.float32(1.0 / float(image_height) * max_num_pixels_to_shift)
    n = len(xmin)
    xmin = xmin - rng.uniform(0, mw, n).astype(np.float32)
    xmax = xmax + rng.uniform(0, mw, n).astype(np.float32)
    ymin = ymin - rng.uniform(0, mh, n).astype(np.float32)
    ymax = ymax + rng.uniform(0, mh, n).astype(np.float32)
    c = lambda v: np.clip(v, np.float32(0.0), np.float32(1.0))
    return c(xmin), c(ymin), c(xmax), c(ymax)


def sample_distorted_bounding_box(height, width, boxes_yxyx, min_object_covered, aspect_ratio_range, area_range,
                                  max_attempts, rng, use_image_if_no_bounding_boxes=True):
    """The algorithm of tf.image.sample_distorted_bounding_box (TF 0.11 kernel): up to max_attempts times draw an
    aspect ratio, a height between the ones that give the minimum / maximum area, a position; accept the first
    crop that contains at least min_object_covered of ANY box; otherwise the whole image.
    Returns (y, x, crop_h, crop_w) in pixels.  boxes: [n, 4] normalised (ymin, xmin, ymax, xmax)."""
    rects = [(int(b[1] * width), int(b[0] * height), int(b[3] * width), int(b[2] * height)) for b in boxes_yxyx]
    if not rects and use_image_if_no_bounding_boxes:
        rects = [(0, 0, width, height)]
    rint = lambda v: int(np.rint(np.float32(v)))
    for _ in range(int(max_attempts)):
        ar = rng.uniform(aspect_ratio_range[0], aspect_ratio_range[1])
        min_area, max_area = area_range[0] * width * height, area_range[1] * width * height
        if max_area <= 0 or ar <= 0:
            continue
        h = rint(np.sqrt(min_area / ar))
        max_h = rint(np.sqrt(max_area / ar))
        if rint(max_h * ar) > width:
            max_h = int((width + 0.5 - 1e-7) / ar)
        max_h = min(max_h, height)
        h = min(h, max_h)
        if h < max_h:
            h += int(rng.randint(0, max_h - h + 1))
        w = rint(h * ar)
        if w * h < min_area:
            h += 1
            w = rint(h * ar)
        area = w * h
        if area < min_area or area > max_area or w > width or h > height or w <= 0 or h <= 0:
            continue
        y = int(rng.randint(0, height - h)) if h < height else 0
        x = int(rng.randint(0, width - w)) if w < width else 0
        for (bx0, by0, bx1, by1) in rects:
            iw, ih = min(bx1, x + w) - max(bx0, x), min(by1, y + h) - max(by0, y)
            inter = max(iw, 0) * max(ih, 0)
            if (bx1 - bx0) * (by1 - by0) * min_object_covered <= inter:
                return y, x, h, w
    return 0, 0, height, width


def crop_bboxes(xmin, ymin, xmax, ymax, image_height, image_width, crop, minimum_area):
    """The box arithmetic of distorted_bounding_box_crop (inputs.py:128-180): boxes in pixels of the FEATURE image
    size, clipped to the crop, shifted to its origin, clipped to [0, image size] (the reference clips to the image,
    not the crop), boxes of area <= minimum_area px dropped, renormalised by the crop size."""
    y0, x0, ch, cw = [np.float32(v) for v in crop]
    H, W = np.float32(image_height), np.float32(image_width)
    sy0 = np.clip(np.maximum(ymin * H, y0) - y0, 0, H)
    sx0 = np.clip(np.maximum(xmin * W, x0) - x0, 0, W)
    sy1 = np.clip(np.minimum(ymax * H, y0 + ch) - y0, 0, H)
    sx1 = np.clip(np.minimum(xmax * W, x0 + cw) - x0, 0, W)
    keep = (sx1 - sx0) * (sy1 - sy0) > np.float32(minimum_area)
    return (sx0[keep] / cw).astype(np.float32), (sy0[keep] / ch).astype(np.float32), \
           (sx1[keep] / cw).astype(np.float32), (sy1[keep] / ch).astype(np.float32)


# ------------------------------------------------------------------------------- colour distortion
def _rgb_to_hsv(rgb):
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    mx, mn = rgb.max(-1), rgb.min(-1)
    d = mx - mn
    s = np.where(mx > 0, d / np.where(mx > 0, mx, 1), 0)
    dz = np.where(d > 0, d, 1)
    h = np.where(mx == r, (g - b) / dz, np.where(mx == g, 2.0 + (b - r) / dz, 4.0 + (r - g) / dz))
    h = np.where(d > 0, (h / 6.0) % 1.0, 0.0)
    return np.stack([h, s, mx], -1)


def _hsv_to_rgb(hsv):
    h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    dh = h * 6.0
    dr = np.clip(np.abs(dh - 3.0) - 1.0, 0, 1)
    dg = np.clip(2.0 - np.abs(dh - 2.0), 0, 1)
    db = np.clip(2.0 - np.abs(dh - 4.0), 0, 1)
    return np.stack([(1 - s + s * dr) * v, (1 - s + s * dg) * v, (1 - s + s * db) * v], -1)


COLOR_BRIGHTNESS, COLOR_SATURATION, COLOR_HUE, COLOR_CONTRAST = 0, 1, 2, 3
_COLOR_ORDERS = {0: (COLOR_BRIGHTNESS, COLOR_SATURATION, COLOR_HUE, COLOR_CONTRAST),          # inputs.py:44-98
                 1: (COLOR_SATURATION, COLOR_BRIGHTNESS, COLOR_CONTRAST, COLOR_HUE),
                 2: (COLOR_CONTRAST, COLOR_HUE, COLOR_BRIGHTNESS, COLOR_SATURATION),
                 3: (COLOR_HUE, COLOR_SATURATION, COLOR_CONTRAST, COLOR_BRIGHTNESS)}
_COLOR_RANGES = {COLOR_BRIGHTNESS: (-32.0 / 255.0, 32.0 / 255.0), COLOR_SATURATION: (0.5, 1.5),
                 COLOR_HUE: (-0.2, 0.2), COLOR_CONTRAST: (0.5, 1.5)}


def color_ops(color_ordering, fast_mode, rng):
    """The (op, argument) list of one distort_color call (inputs.py:44-98): tf.image.random_{brightness,saturation,hue,
    contrast} each draw their delta / factor when applied, so the draws follow the application order."""
    if fast_mode:
        order = (COLOR_BRIGHTNESS, COLOR_SATURATION) if color_ordering == 0 else (COLOR_SATURATION, COLOR_BRIGHTNESS)
    else:
        order = _COLOR_ORDERS.get(color_ordering)
        if order is None:
            raise ValueError("color_ordering must be in [0, 3]")
    return [(op, float(rng.uniform(*_COLOR_RANGES[op]))) for op in order]


def apply_color_ops(image, ops):
    """tf.image.adjust_{brightness,saturation,hue,contrast} restated in float64, applied in order; clipped to [0,1]."""
    img = np.asarray(image, np.float32).astype(np.float64)
    for op, arg in ops:
        if op == COLOR_BRIGHTNESS:
            img = img + arg
        elif op == COLOR_SATURATION:
            hsv = _rgb_to_hsv(img)
            hsv[..., 1] = np.clip(hsv[..., 1] * arg, 0, 1)
            img = _hsv_to_rgb(hsv)
        elif op == COLOR_HUE:
            hsv = _rgb_to_hsv(img)
            hsv[..., 0] = (hsv[..., 0] + arg) % 1.0
            img = _hsv_to_rgb(hsv)
        else:
            m = img.mean((0, 1), keepdims=True)
            img = (img - m) * arg + m
    return np.clip(img, 0.0, 1.0).astype(np.float32)


def distort_color(image, color_ordering, fast_mode, rng):
    """inputs.py:44-98 with tf.image.random_{brightness,saturation,hue,contrast} restated; result clipped to [0,1]."""
    return apply_color_ops(image, color_ops(color_ordering, fast_mode, rng))


class AugmentPlan:
    """Every random decision of one example's augmentation (inputs.py:264-327), drawn in the reference's order, apart
    from the pixels: `crop` (y, x, h, w) in source pixels or None, `method` (tf.image.ResizeMethod 0..3), `color`
    [(op, argument)...] (empty = no colour distortion), `flip`, and the boxes after shift / crop / flip."""
    __slots__ = ("crop", "method", "color", "flip", "xmin", "ymin", "xmax", "ymax")


def plan_augmentation(H, W, image_height, image_width, xmin, ymin, xmax, ymax, cfg, rng):
    """Draw an AugmentPlan for a decoded image of H x W pixels (image_height / image_width are the record's fields the
    box arithmetic uses, inputs.py:264-285).  The pixel work is apply_plan() on the host or mbx_augment_batch on the GPU."""
    p = AugmentPlan()
    n = len(xmin)
    if rng.uniform() < float(cfg.get("DO_RANDOM_BBOX_SHIFT", 0) or 0) and n > 0:                 # inputs.py:264-270
        xmin, ymin, xmax, ymax = shift_bboxes(xmin, ymin, xmax, ymax, image_height, image_width,
                                              cfg.RANDOM_BBOX_SHIFT_EXTENT, rng)
    p.crop = None
    if rng.uniform() < float(cfg.get("DO_RANDOM_CROP", 0) or 0):                                 # inputs.py:272-285
        boxes = np.stack([ymin, xmin, ymax, xmax], 1) if n > 0 else np.zeros((0, 4), np.float32)
        p.crop = tuple(int(v) for v in sample_distorted_bounding_box(
            H, W, boxes, cfg.RANDOM_CROP_MIN_OBJECT_COVERED, cfg.RANDOM_CROP_ASPECT_RATIO_RANGE,
            cfg.RANDOM_CROP_AREA_RANGE, cfg.RANDOM_CROP_MAX_ATTEMPTS, rng))
        xmin, ymin, xmax, ymax = crop_bboxes(xmin, ymin, xmax, ymax, image_height, image_width, p.crop,
                                             cfg.RANDOM_CROP_MINIMUM_AREA)
    p.method = int(rng.randint(0, 4))                                                            # inputs.py:296-302
    do_color = rng.uniform() < float(cfg.get("DO_COLOR_DISTORTION", 0) or 0)                     # inputs.py:312-320
    fast = bool(cfg.get("COLOR_DISTORT_FAST", True))
    ordering = 0 if fast else int(rng.randint(0, 4))
    p.color = color_ops(ordering, fast, rng) if do_color else []
    p.flip = bool(cfg.get("DO_RANDOM_FLIP_LEFT_RIGHT", False) and rng.uniform() < 0.5)           # inputs.py:323-327
    if p.flip:
        xmin, xmax = np.float32(1.0) - xmax, np.float32(1.0) - xmin
    p.xmin, p.ymin, p.xmax, p.ymax = xmin, ymin, xmax, ymax
    return p


def crop_pixels(img, plan):
    if plan.crop is None:
        return img
    y, x, ch, cw = plan.crop
    return img[y:y + ch, x:x + cw]


def apply_plan(image01, plan, S):
    """The pixel half of the augmentation on the host: crop, resize with the drawn method, colour ops, flip.
    image01 is the decoded image in [0,1]; returns [S,S,3] float32 in [0,1]."""
    img = RESIZE_METHODS[plan.method](crop_pixels(image01, plan), S, S)
    if plan.color:
        img = apply_color_ops(img, plan.color)
    if plan.flip:
        img = img[:, ::-1]
    return np.ascontiguousarray(img, np.float32)


def augment_example(image01, image_height, image_width, xmin, ymin, xmax, ymax, cfg, rng):
    """One example through inputs.py:264-327 (the draws in the reference's order): returns
    (image [S,S,3] in [0,1], xmin, ymin, xmax, ymax) with the boxes that survived the crop."""
    p = plan_augmentation(image01.shape[0], image01.shape[1], image_height, image_width, xmin, ymin, xmax, ymax, cfg, rng)
    return apply_plan(image01, p, int(cfg.INPUT_SIZE)), p.xmin, p.ymin, p.xmax, p.ymax


def _records_sharded(tfrecords, shard):
    """Records number i (counted over all files, every epoch alike) with i % n == k for shard = (k, n); the others are
    skipped unparsed (the TFRecord framing gives their length)."""
    k, n = shard
    i = 0
    for path in tfrecords:
        for payload in tfrecord.read_records(path):
            if i % n == k:
                yield tfrecord.parse_example(payload)
            i += 1


def decode_image_u8(jpeg_bytes):
    """tf.image.decode_jpeg(channels=3): [H,W,3] uint8."""
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(jpeg_bytes)).convert("RGB"), dtype=np.uint8)


def train_examples(tfrecords, cfg, max_num_bboxes, num_epochs=None, seed=0, shuffle=False, capacity=1000,
                   min_after_dequeue=96, shard=(0, 1), device_augment=False):
    """Yield prepared training examples (image [S,S,3] float32 in [-1,1], bboxes [G,4] x1,y1,x2,y2, num_bboxes, image_id)
    of this shard of the records: decode + augmentation (inputs.py:200-351), then an optional shuffle pool
    (tf.train.shuffle_batch-like: a random pick once more than min_after_dequeue examples wait).
    device_augment=True leaves the pixel work to the GPU (mbx_augment_batch): the first element is then
    (cropped uint8 pixels [h,w,3], AugmentPlan) -- same draws, same boxes, same order of examples."""
    rng = np.random.RandomState(seed)
    pool = []

    def examples():
        epoch = 0
        while num_epochs is None or epoch < num_epochs:
            got = False
            for ex in _records_sharded(tfrecords, shard):
                got = True
                yield ex
            if not got:
                return
            epoch += 1

    def prepared():
        for ex in examples():
            n = int(ex["image/object/bbox/count"][0])
            f = lambda k: np.array(ex.get(k, []), np.float32)[:n]
            u8 = decode_image_u8(ex["image/encoded"][0])
            p = plan_augmentation(u8.shape[0], u8.shape[1], int(ex["image/height"][0]), int(ex["image/width"][0]),
                                  f("image/object/bbox/xmin"), f("image/object/bbox/ymin"), f("image/object/bbox/xmax"),
                                  f("image/object/bbox/ymax"), cfg, rng)
            bb = np.zeros((max_num_bboxes, 4), np.float32)                                    # inputs.py:338-348
            n = min(len(p.xmin), max_num_bboxes)
            if n > 0:
                bb[:n] = np.stack([p.xmin, p.ymin, p.xmax, p.ymax], 1)[:n]
            image_id = ex["image/id"][0].decode("utf-8")
            if device_augment:
                yield (np.ascontiguousarray(crop_pixels(u8, p)), p), bb, n, image_id
                continue
            img = apply_plan(u8.astype(np.float32) * np.float32(1.0 / 255.0), p, int(cfg.INPUT_SIZE))
            yield (img - np.float32(0.5)) * np.float32(2.0), bb, n, image_id                 # inputs.py:350-351

    def shuffled():
        for item in prepared():
            pool.append(item)
            if len(pool) > max(int(min_after_dequeue), 0) or len(pool) >= int(capacity):
                yield pool.pop(int(rng.randint(0, len(pool))))
        while pool:
            yield pool.pop(int(rng.randint(0, len(pool))))
    return shuffled() if shuffle else prepared()


def train_batches(tfrecords, cfg, batch_size, max_num_bboxes, num_epochs=None, seed=0, shuffle=False, capacity=1000,
                  min_after_dequeue=96):
    """Yield (images [B,S,S,3] in [-1,1], bboxes [B,G,4] x1,y1,x2,y2, num_bboxes [B] int32, image_ids): the
    single-process form; multibox_amd/input_workers.py runs the same example stream in NUM_INPUT_THREADS processes."""
    imgs, boxes, nums, ids = [], [], [], []
    for img, bb, n, image_id in train_examples(tfrecords, cfg, max_num_bboxes, num_epochs, seed, shuffle, capacity,
                                               min_after_dequeue):
        imgs.append(img); boxes.append(bb); nums.append(n); ids.append(image_id)
        if len(imgs) == batch_size:
            yield np.stack(imgs).astype(np.float32), np.stack(boxes), np.array(nums, np.int32), ids
            imgs, boxes, nums, ids = [], [], [], []


def eval_batches(tfrecords, cfg, batch_size, max_num_bboxes, device_images=False):
    """eval_inputs.input_nodes (eval_inputs.py:20-115): one epoch, no augmentation -- decode, legacy bilinear resize to
    INPUT_SIZE, boxes and their original-image areas ('image/object/area') padded to MAX_NUM_BBOXES, [-1,1] scaling;
    the incomplete last batch is dropped like tf.train.(shuffle_)batch at the end of the epoch.
    Yields (images [B,S,S,3], bboxes [B,G,4] x1,y1,x2,y2, num_bboxes [B], areas [B,G], image_ids); with device_images
    the first element is the list of decoded uint8 images instead (resize + scaling then run on the GPU)."""
    S = int(cfg.INPUT_SIZE)
    imgs, boxes, nums, areas, ids = [], [], [], [], []
    for ex in _records(tfrecords):
        n = min(int(ex["image/object/bbox/count"][0]), max_num_bboxes)
        f = lambda k: np.array(ex.get(k, []), np.float32)[:n]
        bb = np.zeros((max_num_bboxes, 4), np.float32)
        ar = np.zeros((max_num_bboxes,), np.float32)
        if n > 0:
            bb[:n] = np.stack([f("image/object/bbox/xmin"), f("image/object/bbox/ymin"), f("image/object/bbox/xmax"),
                               f("image/object/bbox/ymax")], 1)
            a = f("image/object/area")
            ar[:len(a)] = a
        if device_images:                                   # the GPU resizes and scales (augment.BatchAugmenter, method 0)
            imgs.append(decode_image_u8(ex["image/encoded"][0]))
        else:
            img = resize_bilinear_tf(decode_image(ex["image/encoded"][0]), S, S)
            imgs.append((img - np.float32(0.5)) * np.float32(2.0))
        boxes.append(bb); nums.append(n); areas.append(ar); ids.append(ex["image/id"][0].decode("utf-8"))
        if len(imgs) == batch_size:
            yield (imgs if device_images else np.stack(imgs).astype(np.float32)), np.stack(boxes), np.array(nums, np.int32), \
                np.stack(areas), ids
            imgs, boxes, nums, areas, ids = [], [], [], [], []
