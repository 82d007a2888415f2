"""Device side of the training input augmentation: mbx_augment_batch behind a small host class.

The reference runs crop / tf.image.resize_images(random method) / distort_color / random_flip_left_right / (x-0.5)*2
(inputs.py:264-351) as TF CPU ops inside NUM_INPUT_THREADS queue runners.  Here the host (inputs.plan_augmentation)
keeps the random draws, the box arithmetic and the JPEG decode; the pixel work of a whole batch is three HIP launches.
`BatchAugmenter.add()` packs the cropped uint8 pictures of a batch into one pinned buffer (one H2D copy) and
`run()` launches the kernels on the current stream.  There is no host fallback here: without libmbx this raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .inputs import COLOR_CONTRAST

ITEM_DTYPE = np.dtype([("src_offset", "<u8"), ("src_h", "<i4"), ("src_w", "<i4"), ("method", "<i4"), ("flip", "<i4"),
                       ("n_ops", "<i4"), ("op", "<i4", (4,)), ("pad_", "<i4"), ("arg", "<f8", (4,))])
assert ITEM_DTYPE.itemsize == 80                       # mbx_augment_item (include/mbx.h)
METHOD_PREPARED = 4
ALIGN = 16


def source_bytes(h, w):
    return (int(h) * int(w) * 3 + ALIGN - 1) // ALIGN * ALIGN


def fill_item(item, offset, h, w, method, flip, color):
    """Write one mbx_augment_item; `color` is [(op, argument), ...] in application order (inputs.color_ops)."""
    if not 0 <= int(method) <= METHOD_PREPARED or h <= 0 or w <= 0 or len(color) > 4:
        raise ValueError("bad augmentation item: method %r, %rx%r, %d colour ops" % (method, h, w, len(color)))
    item["src_offset"], item["src_h"], item["src_w"] = offset, h, w
    item["method"], item["flip"], item["n_ops"] = method, int(bool(flip)), len(color)
    item["op"][:] = 0
    item["arg"][:] = 0.0
    for k, (op, arg) in enumerate(color):
        if not 0 <= int(op) <= 3:
            raise ValueError("unknown colour op %r" % (op,))
        item["op"][k], item["arg"][k] = op, arg
    return any(op == COLOR_CONTRAST for op, _ in color)


class BatchAugmenter:
    """Buffers for one in-flight batch: pinned staging (pixels + items), their device copies, the kernel workspace and
    the [B,S,S,3] float32 output in [-1,1]."""

    def __init__(self, batch_size, input_size, slot_bytes, device="cuda"):
        import torch
        self.torch = torch
        self.B, self.S = int(batch_size), int(input_size)
        self.slot_bytes = max(int(slot_bytes), self.S * self.S * 3 * 4)
        self.capacity = self.B * (self.slot_bytes + ALIGN)
        self.h_pix = torch.empty((self.capacity,), dtype=torch.uint8, pin_memory=True)
        self.h_items = torch.empty((self.B * ITEM_DTYPE.itemsize,), dtype=torch.uint8, pin_memory=True)
        self.items = self.h_items.numpy().view(ITEM_DTYPE)
        self.d_pix = torch.empty((self.capacity,), dtype=torch.uint8, device=device)
        self.d_items = torch.empty_like(self.h_items, device=device)
        ws = int(_lib.lib().mbx_augment_workspace_bytes(self.B, self.S))
        self.workspace = torch.empty((ws,), dtype=torch.uint8, device=device)
        self.out = torch.empty((self.B, self.S, self.S, 3), dtype=torch.float32, device=device)
        self.used = 0
        self.any_contrast = False
        self.count = 0

    def begin(self):
        self.used, self.any_contrast, self.count = 0, False, 0

    def add(self, pixels, method, flip, color):
        """Append one picture: uint8 [h,w,3] (already cropped) for methods 0..3, or float32 [S,S,3] in [0,1] with
        method 4 (prepared on the host, e.g. a crop too large for a slot)."""
        i = self.count
        if i >= self.B:
            raise ValueError("batch is full")
        pixels = np.ascontiguousarray(pixels)
        if int(method) == METHOD_PREPARED:
            if pixels.dtype != np.float32 or pixels.shape != (self.S, self.S, 3):
                raise ValueError("a prepared picture is float32 [S,S,3]")
            h = w = self.S
        else:
            if pixels.dtype != np.uint8 or pixels.ndim != 3 or pixels.shape[2] != 3:
                raise ValueError("a source picture is uint8 [h,w,3]")
            h, w = pixels.shape[:2]
        raw = pixels.reshape(-1).view(np.uint8)
        if self.used + raw.size > self.capacity:                 # rare: pictures larger than the slots planned for
            torch = self.torch
            torch.cuda.synchronize()                             # earlier launches may still read the old buffers
            self.capacity = max(2 * self.capacity, self.used + raw.size + ALIGN)
            grown = torch.empty((self.capacity,), dtype=torch.uint8, pin_memory=True)
            grown[:self.used] = self.h_pix[:self.used]
            self.h_pix = grown
            self.d_pix = torch.empty((self.capacity,), dtype=torch.uint8, device=self.d_pix.device)
        self.h_pix.numpy()[self.used:self.used + raw.size] = raw
        self.any_contrast |= fill_item(self.items[i], self.used, h, w, method, flip, color)
        self.used += (raw.size + ALIGN - 1) // ALIGN * ALIGN
        self.count += 1

    def upload(self):
        """The H2D copies of the staged pixels and items on the CURRENT stream."""
        if self.count:
            self.d_pix[:self.used].copy_(self.h_pix[:self.used], non_blocking=True)
            self.d_items.copy_(self.h_items, non_blocking=True)

    def launch(self):
        """The three kernel launches on the CURRENT stream (after upload(), ordered by the caller if that ran on another
        stream); returns the device tensor [count,S,S,3]."""
        n = self.count
        if n == 0:
            return self.out[:0]
        _lib.check(_lib.lib().mbx_augment_batch(self.d_pix.data_ptr(), self.d_items.data_ptr(), n, self.S,
                                                int(self.any_contrast), self.out.data_ptr(), self.workspace.data_ptr(),
                                                self.torch.cuda.current_stream().cuda_stream), "mbx_augment_batch")
        return self.out[:n]

    def run(self):
        """upload() + launch() on the CURRENT stream."""
        self.upload()
        return self.launch()


PATCH_DTYPE = np.dtype([("src_offset", "<u8"), ("img_h", "<i4"), ("img_w", "<i4"), ("win_y", "<i4"), ("win_x", "<i4"),
                        ("win_h", "<i4"), ("win_w", "<i4"), ("flip_source", "<i4"), ("pad_", "<i4")])
assert PATCH_DTYPE.itemsize == 40                      # mbx_patch_item (include/mbx.h)


class StagedPatches:
    """One batch laid out for mbx_extract_patches in PINNED host buffers (PatchExtractor.prepare): the uint8 pixels of the
    images it touches, its item table, the number of real patches and the bytes used."""
    __slots__ = ("h_pix", "h_items", "items", "n", "used", "event")

    def __init__(self, torch, B, capacity):
        self.h_pix = torch.empty((int(capacity),), dtype=torch.uint8, pin_memory=True)
        self.h_items = torch.empty((B * PATCH_DTYPE.itemsize,), dtype=torch.uint8, pin_memory=True)
        self.items = self.h_items.numpy().view(PATCH_DTYPE)
        self.n = self.used = 0
        self.event = None                    # recorded behind the upload that last read these buffers


class PatchExtractor:
    """Detection input on the device (row F3): the decoded uint8 images of one batch are uploaded once, every patch
    (original / flipped original / sliding-window crop, detect.py:183-281) is one item of mbx_extract_patches.
    Padding entries (None) give all-zero pictures, like the host path of inputs.detect_batches.

    Two halves (round 4): prepare() -- the HOST work: 24 MB of pixels per batch of 256 patches copied into a pinned staging
    buffer, the item table packed -- may run in the input producer thread, a few batches ahead (a ring of `slots` staging
    buffers); launch() -- two async uploads and one kernel launch -- is all the thread that feeds the GPU does.  In
    detect.py the copy was 9 ms of the main thread's 17.6 ms per batch (the forward pass takes 14)."""

    def __init__(self, batch_size, input_size, device="cuda", capacity_bytes=64 << 20, slots=2):
        import queue
        import torch
        self.torch = torch
        self.B, self.S, self.device = int(batch_size), int(input_size), device
        self.capacity = int(capacity_bytes)
        self.d_items = torch.empty((self.B * PATCH_DTYPE.itemsize,), dtype=torch.uint8, device=device)
        self.d_pix = torch.empty((self.capacity,), dtype=torch.uint8, device=device)
        self.out = torch.empty((self.B, self.S, self.S, 3), dtype=torch.float32, device=device)
        self._free = queue.Queue()
        for _ in range(max(int(slots), 1)):
            self._free.put(StagedPatches(torch, self.B, self.capacity))

    def prepare(self, sources, patches):
        """sources: list of uint8 [H,W,3]; patches: B entries (source index, (y, x, h, w), flip_source) or None.
        Host work only (any thread): returns a StagedPatches for launch().  Blocks while every staging slot is in use."""
        if len(patches) != self.B:
            raise ValueError("expected %d patches, got %d" % (self.B, len(patches)))
        offsets, used = [], 0
        for u8 in sources:
            if u8.dtype != np.uint8 or u8.ndim != 3 or u8.shape[2] != 3:
                raise ValueError("a source picture is uint8 [H,W,3]")
            offsets.append(used)
            used += source_bytes(u8.shape[0], u8.shape[1])
        st = self._free.get()
        if st.event is not None:
            st.event.synchronize()                              # the upload that last read this slot has finished
            st.event = None
        if used > st.h_pix.numel():
            st.h_pix = self.torch.empty((used * 2,), dtype=self.torch.uint8, pin_memory=True)
        hp = st.h_pix.numpy()
        for u8, off in zip(sources, offsets):
            hp[off:off + u8.size] = np.ascontiguousarray(u8).reshape(-1)
        n = next((i for i, p in enumerate(patches) if p is None), len(patches))      # padding entries come last
        if any(p is not None for p in patches[n:]):
            self._free.put(st)
            raise ValueError("padding entries must come last")
        if n:                                                   # whole columns at a time (256 patches per batch in detect.py)
            si = np.fromiter((p[0] for p in patches[:n]), np.int64, n)
            win = np.array([p[1] for p in patches[:n]], np.int64).reshape(n, 4)
            y, x, h, w = win[:, 0], win[:, 1], win[:, 2], win[:, 3]
            H = np.array([u8.shape[0] for u8 in sources], np.int64)[si]
            W = np.array([u8.shape[1] for u8 in sources], np.int64)[si]
            bad = ~((0 <= y) & (0 <= x) & (h > 0) & (w > 0) & (y + h <= H) & (x + w <= W))
            if bad.any():
                b = int(np.argmax(bad))
                self._free.put(st)
                raise ValueError("patch window %r outside its %dx%d image" % (tuple(int(v) for v in win[b]), H[b], W[b]))
            it = st.items[:n]
            it["src_offset"], it["img_h"], it["img_w"] = np.array(offsets, np.uint64)[si], H, W
            it["win_y"], it["win_x"], it["win_h"], it["win_w"] = y, x, h, w
            it["flip_source"] = np.fromiter((int(bool(p[2])) for p in patches[:n]), np.int32, n)
        st.n, st.used = n, used
        return st

    def launch(self, st):
        """Upload a prepared batch and extract its patches on the current stream; returns the device tensor [B,S,S,3] in
        [-1,1] (valid until the next launch).  The staging slot goes back to the ring (reused once its upload is done)."""
        torch = self.torch
        n, used = st.n, st.used
        if used > self.capacity:
            torch.cuda.current_stream().synchronize()           # the previous batch may still read the old device buffer
            self.capacity = used * 2
            self.d_pix = torch.empty((self.capacity,), dtype=torch.uint8, device=self.device)
        if n < self.B:
            self.out[n:].zero_()
        if n:
            self.d_pix[:used].copy_(st.h_pix[:used], non_blocking=True)
            self.d_items.copy_(st.h_items, non_blocking=True)
            st.event = torch.cuda.Event()
            st.event.record(torch.cuda.current_stream())
            _lib.check(_lib.lib().mbx_extract_patches(self.d_pix.data_ptr(), self.d_items.data_ptr(), n, self.S,
                                                      self.out.data_ptr(), torch.cuda.current_stream().cuda_stream),
                       "mbx_extract_patches")
        self._free.put(st)
        return self.out

    def __call__(self, sources, patches):
        """prepare() + launch() in the calling thread."""
        return self.launch(self.prepare(sources, patches))
