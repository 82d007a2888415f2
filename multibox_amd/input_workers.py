"""Multi-process training input + asynchronous host-to-device hand-over.

The reference feeds tf.train.shuffle_batch from NUM_INPUT_THREADS queue runners (inputs.py:353-371,
config.yaml.example:94-100): decoding and augmenting a JPEG costs milliseconds of host time, the training step of this
build takes ~17 ms for 64 images, so one host thread starves it by two orders of magnitude.  Here:

  * ``ParallelTrainInput``: NUM_INPUT_THREADS worker PROCESSES (the pipeline is numpy / PIL code: threads would share
    one GIL).  Worker k prepares the records i with i % n == k (multibox_amd.inputs.train_examples, its own RandomState
    and its own share of the shuffle pool) and writes each finished example into a slot of a ring in shared memory (a
    memory-mapped file under TMPDIR: no /dev/shm size limit, no pickling of the 1 MB images); slot numbers travel
    through two small queues.  The parent assembles batches in arrival order, which interleaves the workers.
    Workers are started with the ``forkserver`` method and never touch the GPU (they do not import torch).
  * ``DevicePrefetcher``: a background thread copies each batch into pinned host buffers and issues the H2D copies
    on a side stream, ``depth`` batches ahead; ``next()`` hands back device tensors whose copies the current stream
    has been made to wait for, so ``Trainer.set_batch`` is a device-to-device copy.

  * device augmentation (``device_augment=True``, train.py's default): the workers only decode the JPEG, draw the
    random decisions and cut the crop window (inputs.plan_augmentation); the ring carries uint8 pixels, the
    prefetcher's side stream uploads them and runs mbx_augment_batch (resize with the drawn method, colour ops, flip,
    [-1,1] scaling: csrc/augment.hip).  Same draws, same boxes, same example order as the host path; pixels
    bit-identical for the resize and within 1e-6 after colour ops (tests/test_gpu_augment.py).

No TensorFlow, no torch DataLoader.  Random draws come from numpy (see inputs.py: "parity unpinned" for sampled values);
with one worker, shuffle off and the same seed the example stream equals train_batches() exactly (tests/test_inputs_cpu.py).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import queue
import tempfile
import threading

import numpy as np


def _worker_main(k, n, tfrecords, cfg_dict, max_num_bboxes, num_epochs, seed, shuffle, capacity, min_after_dequeue,
                 ring_path, slots, S, free_q, ready_q, slot_bytes=0):
    """slot_bytes = 0: the worker does the whole augmentation and a slot holds the finished float32 [S,S,3] picture.
    slot_bytes > 0 (device augmentation): a slot holds the decoded, cropped uint8 pixels and the draws travel with the
    slot number; a crop that does not fit a slot is finished on the host and handed over as a prepared picture."""
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):      # one core per worker: no BLAS thread pools
        os.environ.setdefault(v, "1")
    from .config import Cfg
    from .inputs import train_examples, apply_plan
    device_augment = slot_bytes > 0
    if device_augment:
        ring = np.memmap(ring_path, dtype=np.uint8, mode="r+", shape=(slots, slot_bytes))
    else:
        ring = np.memmap(ring_path, dtype=np.float32, mode="r+", shape=(slots, S, S, 3))
    try:
        stream = train_examples(tfrecords, Cfg(cfg_dict), max_num_bboxes, num_epochs, seed + k, shuffle,
                                max(capacity // n, 1), min_after_dequeue // n, shard=(k, n),
                                device_augment=device_augment)
        for img, bb, nb, image_id in stream:
            slot = free_q.get()
            if slot is None:                                   # the parent is shutting down
                return
            if not device_augment:
                ring[slot] = img
                ready_q.put((slot, bb, nb, image_id))
                continue
            u8, plan = img
            if u8.nbytes <= slot_bytes:
                ring[slot, :u8.nbytes] = u8.reshape(-1)
                ready_q.put((slot, bb, nb, image_id, u8.shape[0], u8.shape[1], plan.method, plan.flip, plan.color))
            else:                                              # rare: finish it here (the crop is already applied)
                plan.crop = None
                f = apply_plan(u8.astype(np.float32) * np.float32(1.0 / 255.0), plan, S)
                ring[slot, :f.nbytes] = f.reshape(-1).view(np.uint8)
                ready_q.put((slot, bb, nb, image_id, S, S, 4, False, []))
        ready_q.put(("done", k))
    except BaseException as e:                                 # surface the failure in the parent instead of hanging it
        ready_q.put(("error", "%s: %r" % (type(e).__name__, e)))
        raise


def _plain(d):
    if isinstance(d, dict):
        return {k: _plain(v) for k, v in d.items()}
    if isinstance(d, (list, tuple)):
        return [_plain(v) for v in d]
    return d


class ParallelTrainInput:
    """Iterator of (images [B,S,S,3] float32, bboxes [B,G,4], num_bboxes [B] int32, image_ids) built by worker processes."""

    def __init__(self, tfrecords, cfg, batch_size, max_num_bboxes, num_workers=None, num_epochs=None, seed=0,
                 shuffle=True, capacity=1000, min_after_dequeue=96, ring_batches=4, tmpdir=None, device_augment=False,
                 max_source_pixels=1024 * 1024):
        """device_augment: the workers decode, draw and crop only; next_into() then takes a BatchAugmenter
        (multibox_amd/augment.py) instead of an image array and the GPU does the pixel work.  max_source_pixels bounds a
        ring slot (a larger crop is finished on the host by its worker)."""
        self.B, self.G, self.S = int(batch_size), int(max_num_bboxes), int(cfg.INPUT_SIZE)
        self.n = max(1, int(num_workers if num_workers is not None else cfg.get("NUM_INPUT_THREADS", 4)))
        self.slots = max(int(ring_batches), 2) * self.B
        self.device_augment = bool(device_augment)
        self.slot_bytes = (max(int(max_source_pixels) * 3, self.S * self.S * 3 * 4) + 15) // 16 * 16 if device_augment else 0
        fd, self.ring_path = tempfile.mkstemp(prefix="mbx_input_ring_", suffix=".bin", dir=tmpdir or os.environ.get("TMPDIR"))
        os.ftruncate(fd, self.slots * (self.slot_bytes or self.S * self.S * 3 * 4))
        os.close(fd)
        if device_augment:
            self.ring = np.memmap(self.ring_path, dtype=np.uint8, mode="r+", shape=(self.slots, self.slot_bytes))
        else:
            self.ring = np.memmap(self.ring_path, dtype=np.float32, mode="r+", shape=(self.slots, self.S, self.S, 3))
        ctx = mp.get_context("forkserver")
        self.free_q, self.ready_q = ctx.Queue(), ctx.Queue()
        for s in range(self.slots):
            self.free_q.put(s)
        self.procs = [ctx.Process(target=_worker_main, daemon=True,
                                  args=(k, self.n, list(tfrecords), _plain(cfg), self.G, num_epochs, int(seed), bool(shuffle),
                                        int(capacity), int(min_after_dequeue), self.ring_path, self.slots, self.S,
                                        self.free_q, self.ready_q, self.slot_bytes)) for k in range(self.n)]
        for p in self.procs:
            p.start()
        self.alive = self.n
        self.closed = False

    def __iter__(self):
        return self

    def __next__(self):
        if self.device_augment:
            raise TypeError("a device_augment input hands out batches through next_into(BatchAugmenter)")
        imgs = np.empty((self.B, self.S, self.S, 3), np.float32)
        self.next_into(imgs)
        return imgs, self._bb, self._n, self._ids

    def next_into(self, images_out):
        """Assemble the next batch with the images written straight into `images_out` (e.g. a pinned buffer; with
        device_augment a BatchAugmenter, which receives the cropped pixels and the draws); returns
        (bboxes, num_bboxes, image_ids).  Raises StopIteration when every worker has finished its epochs."""
        if self.device_augment:
            images_out.begin()
        bb = np.zeros((self.B, self.G, 4), np.float32)
        nn = np.zeros((self.B,), np.int32)
        ids = []
        i = 0
        while i < self.B:
            if self.alive == 0 and self.ready_q.empty():
                raise StopIteration                              # (the incomplete last batch is dropped, like tf.train.batch)
            try:
                item = self.ready_q.get(timeout=1.0)
            except queue.Empty:
                if not any(p.is_alive() for p in self.procs) and self.ready_q.empty():
                    if self.alive:
                        raise RuntimeError("input workers died without reporting")
                    raise StopIteration
                continue
            if item[0] == "done":
                self.alive -= 1
                continue
            if item[0] == "error":
                self.close()
                raise RuntimeError("input worker failed: " + item[1])
            slot, b, n, image_id = item[:4]
            if self.device_augment:
                h, w, method, flip, color = item[4:]
                if method == 4:
                    images_out.add(self.ring[slot, :self.S * self.S * 12].view(np.float32).reshape(self.S, self.S, 3), 4, False, [])
                else:
                    images_out.add(self.ring[slot, :h * w * 3].reshape(h, w, 3), method, flip, color)
            else:
                images_out[i] = self.ring[slot]
            self.free_q.put(slot)
            bb[i], nn[i] = b, n
            ids.append(image_id)
            i += 1
        self._bb, self._n, self._ids = bb, nn, ids
        return bb, nn, ids

    def close(self):
        if self.closed:
            return
        self.closed = True
        for _ in self.procs:
            try:
                self.free_q.put(None)
            except Exception:
                pass
        for p in self.procs:
            p.join(timeout=2.0)
            if p.is_alive():
                p.terminate()                                   # exactly the processes this object started
        try:
            del self.ring
            os.unlink(self.ring_path)
        except OSError:
            pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevicePrefetcher:
    """Pinned double buffers + H2D copies on a side stream, `depth` batches ahead of the training step."""

    def __init__(self, source, batch_size, input_size, max_num_bboxes, device="cuda", depth=2, kernels_on_main=True):
        """With a device_augment source every buffer set owns a BatchAugmenter: the side stream carries the H2D copy of
        the cropped uint8 pixels; the augmentation kernels (mbx_augment_batch, 0.26 ms per batch) are launched by next()
        on the consumer's stream (kernels_on_main, the default) or on the side stream as well."""
        self.kernels_on_main = bool(kernels_on_main)
        import torch
        self.torch, self.src = torch, source
        self.dev = torch.device(device)
        self.depth = max(int(depth), 1)
        B, S, G = batch_size, input_size, max_num_bboxes
        pin = self.dev.type == "cuda"
        self.augment = bool(getattr(source, "device_augment", False))
        if self.augment:
            from .augment import BatchAugmenter
            self.augs = [BatchAugmenter(B, S, source.slot_bytes, device=self.dev) for _ in range(self.depth + 1)]
            self.host = [(torch.empty((B, G, 4), dtype=torch.float32, pin_memory=pin),
                          torch.empty((B,), dtype=torch.int32, pin_memory=pin)) for _ in range(self.depth + 1)]
            self.devb = [(a.out,) + tuple(torch.empty_like(t, device=self.dev) for t in h)
                         for a, h in zip(self.augs, self.host)]
        else:
            self.host = [(torch.empty((B, S, S, 3), dtype=torch.float32, pin_memory=pin),
                          torch.empty((B, G, 4), dtype=torch.float32, pin_memory=pin),
                          torch.empty((B,), dtype=torch.int32, pin_memory=pin)) for _ in range(self.depth + 1)]
            self.devb = [tuple(torch.empty_like(t, device=self.dev) for t in h) for h in self.host]
        self.stream = torch.cuda.Stream(device=self.dev) if pin else None
        self.q = queue.Queue(maxsize=self.depth)
        self.free = queue.Queue()
        for i in range(self.depth + 1):
            self.free.put((i, None))
        self.copied = [None] * (self.depth + 1)                 # per buffer set: event after its last upload
        self.err = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        torch = self.torch
        try:
            while True:
                item = self.free.get()
                if item is None:
                    return
                i, consumed = item
                if self.copied[i] is not None:
                    self.copied[i].synchronize()                # the pinned staging of set i is no longer being read
                if self.augment:
                    hbb, hn = self.host[i]
                    bb, nn, ids = self.src.next_into(self.augs[i])
                else:
                    himg, hbb, hn = self.host[i]
                    if hasattr(self.src, "next_into"):
                        bb, nn, ids = self.src.next_into(himg.numpy())
                    else:
                        images, bb, nn, ids = next(self.src)
                        himg.numpy()[...] = images
                hbb.numpy()[...] = bb
                hn.numpy()[...] = nn
                ev = None
                if self.stream is not None:
                    with torch.cuda.stream(self.stream):
                        if consumed is not None:
                            self.stream.wait_event(consumed)    # the consumer's reads of set i's device buffers are done
                        if self.augment:
                            # the pixel upload rides the side stream; the kernels do too unless kernels_on_main (then
                            # next() launches them in front of the consumer's work)
                            if self.kernels_on_main:
                                self.augs[i].upload()
                            else:
                                self.augs[i].run()
                        for h, d in zip(self.host[i], self.devb[i][-len(self.host[i]):]):
                            d.copy_(h, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                    self.copied[i] = ev
                else:
                    for h, d in zip(self.host[i], self.devb[i][-len(self.host[i]):]):
                        d.copy_(h)
                self.q.put((i, ev, ids))
        except StopIteration:
            self.q.put(None)
        except BaseException as e:
            self.err = e
            self.q.put(None)

    def next(self):
        """(images, bboxes, num_bboxes, image_ids) on the device; the CURRENT stream waits for their copies.  The buffers
        are reused `depth` batches later: consume them (Trainer.set_batch copies) before calling next() that often."""
        item = self.q.get()
        if item is None:
            if self.err is not None:
                raise self.err
            raise StopIteration
        i, ev, ids = item
        if ev is not None:
            self.torch.cuda.current_stream().wait_event(ev)
        if self.augment and self.kernels_on_main and self.stream is not None:
            self.augs[i].launch()
        if getattr(self, "_last", None) is not None:
            # the batch handed out before this one has been consumed -- by work ENQUEUED on the current stream, which the
            # GPU may not have run yet (the host runs ahead of a graph-replayed step): the refill waits for this event
            consumed = None
            if ev is not None:
                consumed = self.torch.cuda.Event()
                consumed.record(self.torch.cuda.current_stream())
            self.free.put((self._last, consumed))
        self._last = i
        return self.devb[i] + (ids,)

    def close(self):
        self.free.put(None)
        if hasattr(self.src, "close"):
            self.src.close()
