"""Data parallelism: one process per GPU, gradients summed over ranks with RCCL
(torch.distributed backend "nccl" = RCCL on ROCm) over xGMI.

The reference is single-device (train.py:184-298); what must be preserved is that its loss is a
SUM over the batch (loss.py:100-101): summing per-rank gradients gives the gradient of the global
batch sum, so the reduction is SUM with no rescaling.  Batch-norm statistics stay per rank
(the reference has no sync-BN to match); rank 0's moving statistics are the ones checkpointed.

Buckets are contiguous slices of the flat fp32 gradient buffer, one per backward segment
(heads -> block8 -> block17 -> ...): each is handed to RCCL as soon as its segment's hipGraph has
been enqueued, so the collective overlaps the rest of backward.  Four buckets of 60 MB (on the
8-GPU xGMI mesh one moves 2*(7/8)*60 MB per GPU: link-bound, not latency-bound), and the parameters
at the bottom of the network (stem, Mixed_5b, block35: 6.7 MB) form a fifth, small bucket -- the only
one nothing overlaps (Trainer._make_segments, tail_params)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class BucketReducer:
    def __init__(self, process_group=None):
        self.pg = process_group
        self.works = []

    @property
    def enabled(self):
        return self.pg is not None and (dist.get_world_size(self.pg) > 1 or bool(os.environ.get("MBX_FORCE_DIST")))

    def reduce_async(self, flat, lo, hi):
        """Start SUM all-reduce of flat[lo:hi] (in place)."""
        if self.enabled and hi > lo and not os.environ.get("MBX_DP_NO_ALLREDUCE"):       # (A/B knob: the form of the step without the collectives)
            self.works.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank` (images / patches are independent units)."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)
