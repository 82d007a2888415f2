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


def rccl_cu_reserve(n_cus=256, env=None):
    """CUs the RCCL kernels of a bucket in flight may hold beside the backward chain: one workgroup per channel.
    NCCL_MAX_NCHANNELS (what a deployment sets to bound it) when given, else RCCL's compiled-in channel ceiling on the
    fully connected 8-GPU xGMI mesh (64: MAXCHANNELS of the gfx94x / gfx950 builds).  Returns (reserve, source)."""
    env = os.environ if env is None else env
    v = env.get("NCCL_MAX_NCHANNELS") or env.get("RCCL_MAX_NCHANNELS")
    if v:
        try:
            n = int(v)
            if n > 0:
                return min(n, n_cus // 2), "NCCL_MAX_NCHANNELS=%d" % n
        except ValueError:
            pass
    return min(64, n_cus // 2), "RCCL default channel ceiling (64)"


def bn_max_workgroups_for(world, n_cus=256, env=None):
    """Grid cap of the launches that hold a GRID BARRIER (one-launch BN backward, fused convolution + BN launches) in a
    data-parallel run: every CU RCCL's channels can occupy is left free, so that the barrier's workgroups stay co-resident
    while a bucket is in flight.  0 (no cap) on one rank.  Returns (cap, {"reserve": ..., "source": ...})."""
    if world <= 1:
        return 0, {"reserve": 0, "source": "single rank"}
    r, src = rccl_cu_reserve(n_cus, env)
    return n_cus - r, {"reserve": r, "source": src}


class BucketReducer:
    def __init__(self, process_group=None):
        self.pg = process_group
        self.works = []
        # A/B knob of tools/dp_ab.sh (the FORM of the bucketed step without its collectives), read ONCE and refused on a real
        # multi-rank group: skipping the all-reduce there would let the ranks diverge silently -- and with it the step
        # control / stop word that rides in the last bucket
        self.no_allreduce = bool(os.environ.get("MBX_DP_NO_ALLREDUCE"))
        if self.no_allreduce and process_group is not None and dist.get_world_size(process_group) > 1:
            raise RuntimeError("MBX_DP_NO_ALLREDUCE is a one-rank A/B knob; refusing it with %d ranks (gradients would never be summed)"
                               % dist.get_world_size(process_group))

    @property
    def enabled(self):
        return self.pg is not None and (dist.get_world_size(self.pg) > 1 or bool(os.environ.get("MBX_FORCE_DIST")))

    def reduce_async(self, flat, lo, hi):
        """Start SUM all-reduce of flat[lo:hi] (in place)."""
        if self.enabled and hi > lo and not self.no_allreduce:
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
