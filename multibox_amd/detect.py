"""Detection post-processing -- host mirror of the reference's detect.py loop on libmbx.

``postprocess`` replaces the per-patch numpy loop detect.py:408-443 with one launch of
``mbx_decode_filter_topk``; ``extract_patches`` keeps the reference's offsets and
restrictions (detect.py:20-72).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib


def extract_patches(image, patch_dims, strides, non_edge_restriction=0.1):
    """detect.py:20-72: same return list [patches, offsets(y,x), restrictions, count]."""
    H, W = image.shape[:2]
    ph, pw = patch_dims
    sh, sw = strides
    hs = list(range(0, H - ph + 1, sh))
    ws = list(range(0, W - pw + 1, sw))
    n = len(hs) * len(ws)
    patches = np.zeros((n, ph, pw, 3), np.float32)
    offs = np.zeros((n, 2), np.int32)
    res = np.zeros((n, 4), np.float32)
    i = 0
    for h in hs:
        for w in ws:
            patches[i] = image[h:h + ph, w:w + pw]
            offs[i] = (h, w)
            res[i] = (0.0 if w == 0 else non_edge_restriction, 0.0 if h == 0 else non_edge_restriction,
                      1.0 if w + pw == W else 1.0 - non_edge_restriction,
                      1.0 if h + ph == H else 1.0 - non_edge_restriction)
            i += 1
    return [patches, offs, res, np.int32(n)]


PATCH_META_DTYPE = np.dtype([("offset_y", "<i4"), ("offset_x", "<i4"), ("patch_h", "<i4"), ("patch_w", "<i4"),
                             ("image_h", "<i4"), ("image_w", "<i4"), ("is_flipped", "<i4"), ("max_to_keep", "<i4"),
                             ("restrictions", "<f4", (4,))])
assert PATCH_META_DTYPE.itemsize == ctypes.sizeof(_lib.PatchMeta)      # mbx_patch_meta (include/mbx.h)


def make_patch_meta(offsets, dims, is_flipped, restrictions, max_to_keep, image_hw, device="cuda"):
    """Pack the per-patch columns fetched at detect.py:398-406 into mbx_patch_meta[B] on the device (whole columns at a
    time: a Python loop over 256 patches was 1.5 ms of the detect loop's main thread)."""
    B = len(offsets)
    m = np.zeros(B, PATCH_META_DTYPE)
    off, dm, hw = np.asarray(offsets).reshape(B, 2), np.asarray(dims).reshape(B, 2), np.asarray(image_hw).reshape(B, 2)
    m["offset_y"], m["offset_x"] = off[:, 0], off[:, 1]
    m["patch_h"], m["patch_w"] = dm[:, 0], dm[:, 1]
    m["image_h"], m["image_w"] = hw[:, 0], hw[:, 1]
    m["is_flipped"] = np.asarray(is_flipped).reshape(B, -1)[:, 0]
    m["max_to_keep"] = np.asarray(max_to_keep).reshape(B, -1)[:, 0]
    m["restrictions"] = np.asarray(restrictions, np.float32).reshape(B, 4)
    return torch.from_numpy(m.view(np.uint8).reshape(-1)).to(device)


class DetectPostprocess:
    """Preallocated outputs for B patches x k_max detections (k_max >= every max_to_keep)."""

    def __init__(self, bbox_priors, batch_size, k_max=200, device="cuda", nms_iou=None):
        """nms_iou: None (the reference: no NMS, detect.py:408-443) or an IoU threshold for the optional greedy
        per-patch NMS stage (mbx_nms, row N1)."""
        self.nms_iou = None if nms_iou is None else float(nms_iou)
        self.priors = torch.as_tensor(bbox_priors, dtype=torch.float32).to(device).contiguous()
        self.P, self.B, self.K = self.priors.shape[0], int(batch_size), int(k_max)
        self.boxes = torch.empty((self.B, self.K, 4), dtype=torch.float64, device=device)
        self.scores = torch.empty((self.B, self.K), dtype=torch.float32, device=device)
        self.index = torch.empty((self.B, self.K), dtype=torch.int32, device=device)
        self.count = torch.empty((self.B,), dtype=torch.int32, device=device)

    def __call__(self, raw_locs, confs, meta):
        """raw_locs [B,P,4] f32, confs [B,P] f32 (sigmoid outputs), meta uint8 tensor from make_patch_meta."""
        B, P = self.B, self.P
        assert raw_locs.shape == (B, P, 4) and confs.numel() == B * P and meta.numel() == B * ctypes.sizeof(_lib.PatchMeta)
        assert raw_locs.is_contiguous() and confs.is_contiguous() and raw_locs.dtype == confs.dtype == torch.float32
        _lib.check(_lib.lib().mbx_decode_filter_topk(raw_locs.data_ptr(), confs.data_ptr(), self.priors.data_ptr(),
                                                     meta.data_ptr(), B, P, self.K, self.boxes.data_ptr(),
                                                     self.scores.data_ptr(), self.index.data_ptr(),
                                                     self.count.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "mbx_decode_filter_topk")
        if self.nms_iou is not None:
            _lib.check(_lib.lib().mbx_nms(self.boxes.data_ptr(), self.scores.data_ptr(), self.index.data_ptr(),
                                          self.count.data_ptr(), B, self.K, self.nms_iou,
                                          torch.cuda.current_stream().cuda_stream), "mbx_nms")
        return self.boxes, self.scores, self.index, self.count


def results_to_json_records(boxes, scores, count, image_ids):
    """detect.py:438-443: list of {"image_id", "bbox", "score"} in patch order."""
    as_np = lambda t: t if isinstance(t, np.ndarray) else t.cpu().numpy()
    boxes, scores, count = as_np(boxes), as_np(scores), as_np(count)
    out = []
    for b in range(boxes.shape[0]):
        n = int(count[b])
        image_id = image_ids[b]
        out.extend({"image_id": image_id, "bbox": bb, "score": sc}
                   for bb, sc in zip(boxes[b, :n].tolist(), scores[b, :n].tolist()))
    return out


from .records import results_to_json_text, batch_chunk, records_to_json      # noqa: E402,F401  (torch-free: run in worker processes)


# ----------------------------------------------------------------------------- multi-GPU detect
# Patches are independent (per-patch loop, detect.py:408): ranks take disjoint batches, NO collective on the data
# path; at the end rank 0 concatenates the per-rank result lists into the one JSON of detect.py:458-460.
def shard_batches(batches, rank, world):
    """Yield (global batch index, batch) for the batches this rank owns: batch i goes to rank i % world, so every
    rank keeps the single-process batching (tf.train.batch order, detect.py:283-292) and the merge below can
    restore the single-process order of the results."""
    for i, b in enumerate(batches):
        if i % world == rank:
            yield i, b


def merge_results(per_rank):
    """per_rank: one list per rank of (global batch index, [records]).  Returns the records in batch order --
    exactly what one process would have written."""
    tagged = [t for part in per_rank for t in part]
    tagged.sort(key=lambda t: t[0])
    assert [t[0] for t in tagged] == sorted(set(t[0] for t in tagged)), "a batch was processed by two ranks"
    return [r for _, recs in tagged for r in recs]


def gather_results(local, group=None):
    """Rank 0 gets merge_results() of every rank's [(batch index, records)]; other ranks get None.  Host-side
    object gather (gloo or RCCL-backed group both work); single process: just the merge."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return merge_results([local])
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = [None] * world if rank == 0 else None
    dist.gather_object(local, parts, dst=0, group=group)
    return merge_results(parts) if rank == 0 else None
