"""The Multibox network on libmbx: Inception-ResNet-v2 backbone + detection heads
(reference model.py:6-337) as a static schedule of HIP kernel launches.

Design (MI355X-first, not a tracing compiler):
  * the graph is built ONCE into two Python lists of pre-bound launches (forward, backward);
    a step replays them on the current HIP stream, so the whole step can be captured into a
    hipGraph (torch.cuda.graph) -- no allocation, no host sync inside;
  * activations are NHWC bf16 channel-slice views of a few wide buffers: every tf.concat of
    model.py is free, sibling 1x1 convs that read the same tensor run as ONE GEMM
    (block35: N = 96, block17: 320, block8: 384, Mixed_5b: 208, Mixed_7a: 768);
  * parameters live in flat fp32 buffers (weights+biases | betas | moving stats) so the
    optimizer/EMA step is one launch per range and the data-parallel gradient all-reduce is a
    handful of contiguous buckets.

Variable names follow the slim scopes of the reference (model.py:87,142,202,207) so that a
TF checkpoint maps onto them (multibox_amd/tf_checkpoint.py).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import re

import numpy as np
import torch

from . import _lib, ops
from .ops import View

BN_EPS = 0.001           # train.py:96
STATS_ROWS = int(os.environ.get("MBX_STATS_ROWS", "8"))          # replica rows of the atomically added batch-norm statistics (mbx_conv_desc.stats_rows_mod)
WEIGHT_DECAY = 0.00004   # train.py:104-105


def _same_pad(n, k, s):
    """TF 'SAME': out = ceil(n/s); extra padding goes to the bottom/right."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def _valid(n, k, s):
    return (n - k) // s + 1, 0


class ParamSpec:
    """One reference variable group: conv weights (KRSC) [+ bias] [+ BN beta / moving stats]."""

    def __init__(self, scope, K, R, S, Cin, bn, bias):
        self.scope, self.K, self.R, self.S, self.Cin, self.bn, self.bias = scope, K, R, S, Cin, bn, bias


class ConvOp:
    """A (possibly fused) convolution with its normalisation/activation, forward and backward."""

    def __init__(self, net, members, x: View, out: View, R, S, stride, pad_t, pad_l, kind, relu=True, skip=None,
                 rscale=0.0, trainable=True, need_dx=True, head=None):
        self.net, self.members, self.x, self.out = net, members, x, out
        self.R, self.S, self.stride, self.pad_t, self.pad_l = R, S, stride, pad_t, pad_l
        self.kind, self.relu, self.skip, self.rscale = kind, relu, skip, rscale       # kind: bn | frozen | residual | head
        self.trainable, self.need_dx, self.head = trainable, need_dx, head
        self.K = sum(m.K for m in members)
        self.Cin = x.C
        self.M = out.M
        self.name = members[0].scope if len(members) == 1 else "+".join(m.scope for m in members)

    # offsets into the flat parameter buffers are assigned by Net.finalize()
    w_off = b_off = beta_off = dw_off = -1


class BnGroup:
    """Sibling convolutions normalised by ONE finalize / apply / backward launch (mbx.h, BATCH-NORM GROUPS): their pre-BN
    outputs (and gradients) are channel slices of one contiguous [M, K] tensor, member i at channels koff[i]."""

    def __init__(self, members):
        self.members = list(members)
        self.K = sum(m.K for m in members)
        self.koff, k = [], 0
        for m in members:
            self.koff.append(k)
            k += m.K
        self.y = self.dy = None
        self.fwd_desc, self.dgrad_desc = {}, {}   # member index -> its forward / data-gradient descriptor
        self.pair_fwd = self.pair_bwd = False     # a group of TWO: both convolutions in one launch (mbx_conv_pair), if it applies
        self.rows = [0] * len(members)          # statistics partial rows of each member's convolution (tile height)
        self.stats_off = [0] * len(members)     # float offsets of the members' partial rows in Net.stats_scratch

    def chan_map(self, views):
        """(base channel, mbx_chan_map) placing member i's channels at views[i].ch_off of the common buffer."""
        rel = [v.ch_off - ko for v, ko in zip(views, self.koff)]
        base = min(rel)
        m = _lib.ChanMap()
        m.n = len(self.members)
        for i, (ko, r) in enumerate(zip(self.koff, rel)):
            m.c_begin[i], m.offset[i] = ko, r - base
        return base, m


class Net:
    def __init__(self, batch, input_size=299, k=5, mode="train", fine_tune=False, device="cuda", seed=2,
                 bn_decay=0.9997, repeats=(10, 20, 9), bn_max_workgroups=0, wgrad_overlap_cus=None):
        """repeats: number of block35 / block17 / block8 repetitions (model.py:142,162,187); anything
        but the reference's (10, 20, 9) is a reduced-depth network for tests."""
        assert mode in ("train", "infer")
        self.repeats = tuple(repeats)
        self.B, self.S, self.k, self.mode, self.fine_tune, self.dev = batch, input_size, k, mode, fine_tune, device
        self.bn_decay = bn_decay
        self.tune_registry = []                    # (key, descriptor, fwd | dgrad | wgrad) of every tunable conv launch
        # MBX_DETERMINISTIC=1: bit-reproducible gradients for bisecting parity bugs -- three-launch batch-norm backward
        # (no atomics) and un-split weight-gradient tiles (one adder per element); slower, same mathematics
        self.deterministic = bool(int(os.environ.get("MBX_DETERMINISTIC", "0")))
        self.no_onepass = self.deterministic or bool(int(os.environ.get("MBX_NO_BN_ONEPASS", "0")))    # A/B knob: three-launch BN backward
        # Forward batch-norm statistics ADDED by the convolution epilogues into STATS_ROWS replica rows per layer (64-bit
        # fixed-point INTEGER atomics: order-independent, so the forward pass stays bit-reproducible; cleared with the work
        # counters at the start of forward()), reduced by the apply launch itself: no finalize launch.  Measured (round 4,
        # LAB_NOTES): 130 launches of 4.5 us go, the apply launch grows by 2.2 us (the rows prologue) and every statistics
        # convolution by 0.3-0.5 us (the atomics' acknowledgement at the end of the kernel): -0.05..-0.14 ms per step.
        # MBX_ATOMIC_STATS=0: a plain float32 row per tile + finalize (also what MBX_DETERMINISTIC=1 uses).
        self.atomic_stats = mode == "train" and not self.deterministic and os.environ.get("MBX_ATOMIC_STATS", "1") != "0"
        # batch-norm BACKWARD statistics produced by the data gradient that writes the activation gradient (mbx_bn_bwd_stats):
        # the layer's backward then is ONE streaming launch (mbx_bn_bwd_apply_rows) instead of the grid-barrier launch --
        # wherever every channel of the layer's activation has exactly one consumer and that is a stride-1 convolution
        # (_plan_bw_stats: the branch layers of block35 / block17 / block8, 121 of 141 launches).  Needs the relu
        # thresholds the atomic-statistics forward writes.  OPTION, off (MBX_BW_STATS=1).  Measured (round
        # 4): the backward launch drops from 14.2 to 8.9 us, but the statistics epilogue costs its data gradient 3.3-5.5 us
        # (threshold loads, sums, cross-wave reduce, the atomics' acknowledgement -- all on the tail of a launch that has one
        # tile per CU) and 17 us where the 256 x 128 tile has no room for it: 15.98 -> 16.5 ms per step.
        self.bw_stats = self.atomic_stats and os.environ.get("MBX_BW_STATS", "0") == "1"
        # ROUND 6: the BN apply of a training-mode convolution as the TAIL of the convolution's own launch (mbx_conv_desc.bn_apply,
        # csrc/fused_bn.h): workgroups meet at a one-shot grid barrier, reduce the statistics rows themselves and normalise the
        # tiles they wrote, read back from their own L2 -- no bn_apply_rows launch.  Where the library supports it (persistent
        # igemm5 tiles, the whole-width direct and the resident-image launch; not batch-norm groups, split-K, the stem's fused
        # pools).  Bit-identical to the two-launch form, which stays as the call-time fall-back (no_onepass: a grid barrier timed
        # out).  MEASURED LEVEL, so OFF by default (LAB_NOTES round 6: one box, alternating, 14.70 ms split / 14.78-14.80 fused;
        # per launch a tail costs 5-8 us against 5-9 us for the apply launch it replaces -- waiting for the atomics'
        # acknowledgement, the barrier's two memory-side round trips and the statistics read-back add up to what a kernel
        # boundary in a replayed graph costs).  MBX_FUSE_APPLY=1 selects it: 114 fewer kernels per step.
        self.fuse_apply = self.atomic_stats and os.environ.get("MBX_FUSE_APPLY", "0") == "1"
        self.fused_apply_launches = 0
        # ... and the BN BACKWARD of a layer as the tail of the data-gradient launch(es) that WRITE its activation gradient
        # (mbx_conv_desc.bn_bwd): wherever every channel of a batch-norm layer (group) has exactly one consumer and that is a
        # stride-1 convolution whose data gradient is a persistent / one-tile-per-workgroup launch (_plan_fused_bwd).  The
        # layer's own backward launch (mbx_bn_bwd_onepass) is then not issued.  Not in deterministic mode (float atomics);
        # The three-launch form stays the call-time fall-back (no_onepass).  MEASURED LEVEL as well (14.70 -> 14.82 ms with both
        # fusions, 725 -> 514 kernels; a tail on one tile per workgroup costs 9-15 us against the 9-15 us of the launch(es) it
        # replaces, on several tiles more: _plan_fused_bwd's one-round rule): OFF by default, MBX_FUSE_BWD=1 selects it.
        self.fuse_bwd = (mode == "train" and not self.deterministic and not self.bw_stats and torch.device(device).type == "cuda"
                         and os.environ.get("MBX_FUSE_BWD", "0") == "1")
        self.fused_bwd_launches = self.fused_bwd_layers = 0
        # relu backward of the residual block outputs from SIGN BITS (mbx_conv_desc.relu_bits): the residual launch writes one
        # bit per element beside its bf16 output, the data gradient that applies the mask reads that byte per eight channels
        # instead of 16 bytes of the tensor -- its epilogue streams two tensors and a sixteenth instead of three.
        # MBX_RELU_BITS=0: the bf16 tensor as the mask (rounds 2-3).
        self.relu_bits = mode == "train" and os.environ.get("MBX_RELU_BITS", "1") != "0"
        # every conv launch times the library's tile pick against the other tile configurations once, at build time
        self.autotune = torch.device(device).type == "cuda" and bool(int(os.environ.get("MBX_AUTOTUNE", "1")))
        # grid cap of the one-launch BN backward: data-parallel runs leave CUs to the RCCL kernels of the bucket in flight
        self.bn_max_wg = int(os.environ.get("MBX_BN_MAX_WG", "0")) or (bn_max_workgroups or 0)
        # overlapped weight gradients (ops.WG_OVERLAP_DEFAULT, Trainer._capture): while a capped grouped weight-gradient launch
        # holds `wgrad_overlap_cus` CUs on a second stream, the persistent launches of the backward chain take at most
        # `chain_cap` workgroups -- self.cu_cap, set by the Trainer around the segments it overlaps (0: no cap).  The layers
        # whose one-launch BN backward does not fit `chain_cap` workgroups use the three-launch form.
        self.wgrad_overlap_cus = ops.wgrad_overlap_cus() if wgrad_overlap_cus is None else int(wgrad_overlap_cus)
        self.n_cus = torch.cuda.get_device_properties(device).multi_processor_count if torch.device(device).type == "cuda" else 256
        self.chain_cap = 0
        if self.wgrad_overlap_cus > 0 and mode == "train":
            self.wgrad_overlap_cus = min(self.wgrad_overlap_cus, self.n_cus // 2)
            self.chain_cap = min(self.bn_max_wg or self.n_cus, self.n_cus - self.wgrad_overlap_cus)
        self.cu_cap = 0
        # defer_moving (the Trainer sets it): the forward pass leaves the moving statistics alone -- bn_finalize stores the batch
        # variance beside the batch mean -- and apply_moving_update() applies `moving -= (1-decay)(moving - batch)` for every
        # layer in ONE launch behind the backward pass, gated by the step control word like the optimiser: a step that is
        # skipped (barrier timeout, stop request) is skipped everywhere.  Same float32 expressions: bit-identical values.
        self.defer_moving = False
        # work counters of the persistent igemm5 launches (mbx_conv_desc.work_counter): one per launch, cleared together
        # at the start of every pass (forward() / backward())
        self.i5_counters = torch.zeros(8192, dtype=torch.int32, device=device)
        self._i5_used = 0
        # timing probes of tools/whatif_probe.py (wrong results while set: work is REMOVED to bound what a restructuring could save)
        self._probe_skip_apply = self._probe_skip_bn_bwd = False
        self.convs, self.fwd, self.bwd = [], [], []
        self._bn_group_requests, self.bn_groups = [], []
        self.grad_alias = {}       # id(activation buffer) -> gradient buffer
        self.written = set()       # gradient regions already written in the backward pass (build time)
        self.heads = []
        self._bufs = []
        self._build()
        self._finalize(seed)

    # ------------------------------------------------------------------ buffers
    def alloc(self, H, W, Cc, dtype=torch.bfloat16, zero=False):
        v = View.alloc(self.B, H, W, Cc, dtype=dtype, device=self.dev, zero=zero)
        self._bufs.append(v.buf)
        return v

    def grad_of(self, v: View) -> View:
        """Gradient view mirroring an activation view (same ld / channel offset)."""
        key = id(v.buf)
        if key not in self.grad_alias:
            g = torch.zeros_like(v.buf)
            self._bufs.append(g)
            self.grad_alias[key] = g
        return View(self.grad_alias[key], v.N, v.H, v.W, v.C, v.ld, v.ch_off, v.elem_size)

    def share_grad(self, v: View, like: View):
        """Make v's buffer use the same gradient buffer as `like`'s (identical shapes)."""
        self.grad_alias[id(v.buf)] = self.grad_of(like).buf

    # --------------------------------------------------------------- graph building
    def _training_bn(self, is_head):
        if self.mode == "infer":
            return False
        return is_head if self.fine_tune else True

    def conv(self, members, x, out, kh, kw, stride=1, padding="SAME", kind=None, relu=True, skip=None, rscale=0.0,
             is_head=False, need_dx=True, head=None):
        """members: list of (scope, K).  Creates a ConvOp writing into `out`."""
        fh = _same_pad if padding == "SAME" else _valid
        Ho, pt = fh(x.H, kh, stride)
        Wo, pl = fh(x.W, kw, stride)
        assert (Ho, Wo) == (out.H, out.W), (members, Ho, Wo, out.H, out.W)
        if kind is None:
            kind = "bn" if self._training_bn(is_head) else "frozen"
        trainable = self.mode == "train" and (is_head or not self.fine_tune)
        specs = [ParamSpec(s, K, kh, kw, x.C, bn=kind in ("bn", "frozen"), bias=kind == "residual") for s, K in members]
        op = ConvOp(self, specs, x, out, kh, kw, stride, pt, pl, kind, relu, skip, rscale, trainable,
                    need_dx and trainable, head)
        op.is_head = is_head
        assert op.K == out.C or kind == "head", (op.name, op.K, out.C)
        self.convs.append(op)
        self.fwd.append(op)
        return op

    def group_bn(self, *members):
        """Declare sibling convolutions (consecutive in forward order, same pixels, mutually independent inputs) a batch-
        norm group.  Taken up only where it applies (training-mode BN on all of them, outputs in one buffer; _alloc_scratch);
        MBX_BN_GROUPS=0 turns it off (A/B)."""
        if os.environ.get("MBX_BN_GROUPS", "1") != "0" and len(members) >= 2:
            self._bn_group_requests.append(members)

    def _build(self):
        B, S, k = self.B, self.S, self.k
        P = "InceptionResnetV2/"
        self.images = self.alloc(S, S, 8)                      # packed bf16 input (3 real channels)
        s1, _ = _valid(S, 3, 2)
        a1 = self.alloc(s1, s1, 32)
        self.conv([(P + "Conv2d_1a_3x3", 32)], self.images, a1, 3, 3, 2, "VALID", need_dx=False)
        s2 = s1 - 2
        a2 = self.alloc(s2, s2, 32)
        self.conv([(P + "Conv2d_2a_3x3", 32)], a1, a2, 3, 3, 1, "VALID")
        a3 = self.alloc(s2, s2, 64)
        self.conv([(P + "Conv2d_2b_3x3", 64)], a2, a3, 3, 3, 1, "SAME")
        s3, _ = _valid(s2, 3, 2)
        p3 = self.alloc(s3, s3, 64)
        self.pool("max", a3, p3, 3, 2, scope=P + "MaxPool_3a_3x3")
        a4 = self.alloc(s3, s3, 80)
        self.conv([(P + "Conv2d_3b_1x1", 80)], p3, a4, 1, 1, 1, "VALID")
        s4 = s3 - 2
        a5 = self.alloc(s4, s4, 192)
        self.conv([(P + "Conv2d_4a_3x3", 192)], a4, a5, 3, 3, 1, "VALID")
        s5, _ = _valid(s4, 3, 2)
        t5 = self.alloc(s5, s5, 192)
        self.pool("max", a5, t5, 3, 2, scope=P + "MaxPool_5a_3x3")

        # ---- Mixed_5b (model.py:120-141): [t1 48 | t2 64 | b0 96 | b1 64 | b2 96 | b3 64 | t2b 96]
        Q = P + "Mixed_5b/"
        z = self.alloc(s5, s5, 528)
        self.conv([(Q + "Branch_1/Conv2d_0a_1x1", 48), (Q + "Branch_2/Conv2d_0a_1x1", 64), (Q + "Branch_0/Conv2d_1x1", 96)],
                  t5, z.slice(0, 208), 1, 1)
        ga = self.conv([(Q + "Branch_1/Conv2d_0b_5x5", 64)], z.slice(0, 48), z.slice(208, 64), 5, 5)
        gb = self.conv([(Q + "Branch_2/Conv2d_0b_3x3", 96)], z.slice(48, 64), z.slice(432, 96), 3, 3)
        self.group_bn(ga, gb)
        self.conv([(Q + "Branch_2/Conv2d_0c_3x3", 96)], z.slice(432, 96), z.slice(272, 96), 3, 3)
        p5 = self.alloc(s5, s5, 192)
        self.pool("avg", t5, p5, 3, 1, pad=1, scope=Q + "Branch_3/AvgPool_0a_3x3")
        self.conv([(Q + "Branch_3/Conv2d_0b_1x1", 64)], p5, z.slice(368, 64), 1, 1)
        net = z.slice(112, 320)
        self.endpoints = {"MaxPool_5a_3x3": t5, "Mixed_5b": net}

        # ---- 10 x block35 (model.py:6-24), scale 0.17
        trunk0 = net
        zg = None
        for i in range(1, self.repeats[0] + 1):
            Q = P + "Repeat/block35_%d/" % i
            z = self.alloc(s5, s5, 240)     # [t1 32 | t2 32 | b0 32 | b1 32 | b2 64 | b2a 48]
            if zg is None:
                zg = z
            else:
                self.share_grad(z, zg)
            self.conv([(Q + "Branch_1/Conv2d_0a_1x1", 32), (Q + "Branch_2/Conv2d_0a_1x1", 32), (Q + "Branch_0/Conv2d_1x1", 32)],
                      net, z.slice(0, 96), 1, 1)
            ga = self.conv([(Q + "Branch_1/Conv2d_0b_3x3", 32)], z.slice(0, 32), z.slice(96, 32), 3, 3)
            gb = self.conv([(Q + "Branch_2/Conv2d_0b_3x3", 48)], z.slice(32, 32), z.slice(192, 48), 3, 3)
            self.group_bn(ga, gb)           # independent siblings (model.py:11-17): one finalize / apply / backward launch
            self.conv([(Q + "Branch_2/Conv2d_0c_3x3", 64)], z.slice(192, 48), z.slice(128, 64), 3, 3)
            out = self.alloc(s5, s5, 320)
            self.residual(Q + "Conv2d_1x1", z.slice(64, 128), net, out, 0.17, True, trunk0)
            net = out

        # ---- Mixed_6a (model.py:145-161): [b0 384 | b1 384 | pool 320]
        Q = P + "Mixed_6a/"
        s6, _ = _valid(s5, 3, 2)
        o6 = self.alloc(s6, s6, 1088)
        self.conv([(Q + "Branch_0/Conv2d_1a_3x3", 384)], net, o6.slice(0, 384), 3, 3, 2, "VALID")
        t6a = self.alloc(s5, s5, 256)
        self.conv([(Q + "Branch_1/Conv2d_0a_1x1", 256)], net, t6a, 1, 1)
        t6b = self.alloc(s5, s5, 256)
        self.conv([(Q + "Branch_1/Conv2d_0b_3x3", 256)], t6a, t6b, 3, 3)
        self.conv([(Q + "Branch_1/Conv2d_1a_3x3", 384)], t6b, o6.slice(384, 384), 3, 3, 2, "VALID")
        self.pool("max", net, o6.slice(768, 320), 3, 2, scope=Q + "Branch_2/MaxPool_1a_3x3")
        self.endpoints["block35_10"] = net
        net = o6
        self.endpoints["Mixed_6a"] = net

        # ---- 20 x block17 (model.py:27-44), scale 0.10
        trunk0, zg = net, None
        for i in range(1, self.repeats[1] + 1):
            Q = P + "Repeat_1/block17_%d/" % i
            z = self.alloc(s6, s6, 672)     # [t1 128 | b0 192 | b1_2 192 | b1_1 160]
            if zg is None:
                zg = z
            else:
                self.share_grad(z, zg)
            self.conv([(Q + "Branch_1/Conv2d_0a_1x1", 128), (Q + "Branch_0/Conv2d_1x1", 192)], net, z.slice(0, 320), 1, 1)
            self.conv([(Q + "Branch_1/Conv2d_0b_1x7", 160)], z.slice(0, 128), z.slice(512, 160), 1, 7)
            self.conv([(Q + "Branch_1/Conv2d_0c_7x1", 192)], z.slice(512, 160), z.slice(320, 192), 7, 1)
            out = self.alloc(s6, s6, 1088)
            self.residual(Q + "Conv2d_1x1", z.slice(128, 384), net, out, 0.10, True, trunk0)
            net = out

        # ---- Mixed_7a (model.py:164-185): [b0 384 | b1 288 | b2 320 | pool 1088]
        Q = P + "Mixed_7a/"
        s7, _ = _valid(s6, 3, 2)
        o7 = self.alloc(s7, s7, 2080)
        t7 = self.alloc(s6, s6, 1056)       # [t0 256 | t1 256 | t2 256 | t2b 288]
        self.conv([(Q + "Branch_0/Conv2d_0a_1x1", 256), (Q + "Branch_1/Conv2d_0a_1x1", 256), (Q + "Branch_2/Conv2d_0a_1x1", 256)],
                  net, t7.slice(0, 768), 1, 1)
        ga = self.conv([(Q + "Branch_0/Conv2d_1a_3x3", 384)], t7.slice(0, 256), o7.slice(0, 384), 3, 3, 2, "VALID")
        gb = self.conv([(Q + "Branch_1/Conv2d_1a_3x3", 288)], t7.slice(256, 256), o7.slice(384, 288), 3, 3, 2, "VALID")
        self.group_bn(ga, gb)
        self.conv([(Q + "Branch_2/Conv2d_0b_3x3", 288)], t7.slice(512, 256), t7.slice(768, 288), 3, 3)
        self.conv([(Q + "Branch_2/Conv2d_1a_3x3", 320)], t7.slice(768, 288), o7.slice(672, 320), 3, 3, 2, "VALID")
        self.pool("max", net, o7.slice(992, 1088), 3, 2, scope=Q + "Branch_3/MaxPool_1a_3x3")
        self.endpoints["block17_20"] = net
        net = o7
        self.endpoints["Mixed_7a"] = net

        # ---- 9 x block8 (scale 0.20) + Block8 without relu at scale 1.0 (model.py:187-188)
        trunk0, zg = net, None
        n8 = self.repeats[2] + 1
        for i in range(1, n8 + 1):
            Q = P + ("Repeat_2/block8_%d/" % i if i < n8 else "Block8/")
            z = self.alloc(s7, s7, 864)     # [t1 192 | b0 192 | b1_2 256 | b1_1 224]
            if zg is None:
                zg = z
            else:
                self.share_grad(z, zg)
            self.conv([(Q + "Branch_1/Conv2d_0a_1x1", 192), (Q + "Branch_0/Conv2d_1x1", 192)], net, z.slice(0, 384), 1, 1)
            self.conv([(Q + "Branch_1/Conv2d_0b_1x3", 224)], z.slice(0, 192), z.slice(640, 224), 1, 3)
            self.conv([(Q + "Branch_1/Conv2d_0c_3x1", 256)], z.slice(640, 224), z.slice(384, 256), 3, 1)
            out = self.alloc(s7, s7, 2080)
            self.residual(Q + "Conv2d_1x1", z.slice(192, 448), net, out, 0.20 if i < n8 else 1.0, i < n8, trunk0)
            net = out
        feat = self.alloc(s7, s7, 1536)
        self.conv([(P + "Conv2d_7b_1x1", 1536)], net, feat, 1, 1)
        self.features = feat
        self.endpoints["Conv2d_7b_1x1"] = feat

        # ------------------------------------------------------- detection heads (model.py:198-324)
        H = "Multibox/"
        f = s7
        grids = []
        h8 = self.alloc(f, f, 192)
        self.conv([(H + "8x8/Conv", 96)], feat, h8.slice(0, 96), 1, 1, is_head=True, need_dx=not self.fine_tune)
        self.conv([(H + "8x8/Conv_1", 96)], h8.slice(0, 96), h8.slice(96, 96), 3, 3, is_head=True)
        grids.append((H + "8x8/", h8.slice(96, 96), f, k))
        h6a = self.alloc(f, f, 96)
        self.conv([(H + "6x6/Conv", 96)], feat, h6a, 3, 3, is_head=True, need_dx=not self.fine_tune)
        h6b = self.alloc(f - 2, f - 2, 96)
        self.conv([(H + "6x6/Conv_1", 96)], h6a, h6b, 3, 3, 1, "VALID", is_head=True)
        grids.append((H + "6x6/", h6b, f - 2, k))
        f4 = -(-f // 2)
        n4 = self.alloc(f4, f4, 256)
        self.conv([(H + "Conv", 256)], feat, n4, 3, 3, 2, "SAME", is_head=True, need_dx=not self.fine_tune)
        h432 = self.alloc(f4, f4, 384)      # [4x4/Conv 128 | 3x3/Conv 128 | 2x2/Conv 128]: siblings on n4, one batch-norm group
        h4, h32 = h432.slice(0, 128), h432.slice(128, 256)
        ga = self.conv([(H + "4x4/Conv", 128)], n4, h4, 3, 3, is_head=True)
        grids.append((H + "4x4/", h4, f4, k))
        gb = self.conv([(H + "3x3/Conv", 128), (H + "2x2/Conv", 128)], n4, h32, 1, 1, is_head=True)
        self.group_bn(ga, gb)
        h3 = self.alloc(f4 - 1, f4 - 1, 96)
        self.conv([(H + "3x3/Conv_1", 96)], h32.slice(0, 128), h3, 2, 2, 1, "VALID", is_head=True)
        grids.append((H + "3x3/", h3, f4 - 1, k))
        h2 = self.alloc(f4 - 2, f4 - 2, 96)
        self.conv([(H + "2x2/Conv_1", 96)], h32.slice(128, 128), h2, 3, 3, 1, "VALID", is_head=True)
        grids.append((H + "2x2/", h2, f4 - 2, k))
        f1 = f - 7
        g1 = self.alloc(f1, f1, 1536)
        self.pool("avg", feat, g1, 8, 1, pad=0, scope=H + "1x1/AvgPool2D")
        grids.append((H + "1x1/", g1, f1, 1))
        # prediction index: off_g + (i*g + j)*k + a  (model.py:296-319)
        self.head_ld = max(8, (5 * k + 7) // 8 * 8)       # 5k head channels padded to 8 (k=5: 32, k=7: 40)
        self.P = sum(g * g * kk for _, _, g, kk in grids)
        self.grid_sizes = [g for _, _, g, _ in grids]
        self.locs = torch.zeros((B, self.P, 4), dtype=torch.float32, device=self.dev)
        self.logits = torch.zeros((B, self.P), dtype=torch.float32, device=self.dev)
        self.d_locs = torch.zeros_like(self.locs)
        self.d_logits = torch.zeros_like(self.logits)
        off = 0
        for scope, src, g, kk in grids:
            o = self.alloc(g, g, self.head_ld, dtype=torch.float32, zero=True)   # [M_g, head_ld] f32: 4k locs | k logits
            if scope.endswith("1x1/"):
                names = [(scope + "Conv", 4), (scope + "Conv_1", 1)]
            elif scope.endswith("4x4/"):
                names = [(scope + "Conv_1", 4 * kk), (scope + "Conv_2", kk)]
            else:
                names = [(scope + "Conv_2", 4 * kk), (scope + "Conv_3", kk)]
            op = self.conv(names, src, o, 1, 1, kind="head", relu=False, is_head=True, head=(g * g, kk, off),
                           need_dx=not (self.fine_tune and src.buf is feat.buf))
            self.heads.append(op)
            off += g * g * kk

    def residual(self, scope, mixed, skip, out, scale, relu, trunk0):
        op = self.conv([(scope, out.C)], mixed, out, 1, 1, kind="residual", relu=relu, skip=skip, rscale=scale)
        # the gradient of every trunk tensor of a residual stage lives in ONE buffer, updated in place
        op.trunk0 = trunk0
        return op

    def pool(self, kind, x, out, ksz, stride, pad=0, scope=None):
        op = PoolOp(self, kind, x, out, ksz, stride, pad)
        op.scope = scope          # slim scope of the pooling layer (model.py), for the layer-table parity test
        self.fwd.append(op)

    # ------------------------------------------------------------- parameters
    def _finalize(self, seed):
        dev = self.dev
        # flat parameter layout: W = all filters (+ biases), 8-element aligned; Bt = betas; stats
        w_off = bt_off = 0
        self.param_index = {}
        for op in self.convs:
            op.w_off = w_off
            ktot = op.R * op.S * op.Cin
            row = 0
            for m in op.members:
                # the stem filter is stored with C_in padded 3 -> 8 (zeros); the variable is [K,R,S,3]
                real_c = 3 if op.x is self.images else m.Cin
                self.param_index[m.scope + "/weights"] = ("W", w_off + row * ktot, (m.K, m.R, m.S, real_c), m.Cin)
                row += m.K
            w_off += op.K * ktot
            w_off = (w_off + 7) // 8 * 8
            if op.kind == "residual":
                op.b_off = w_off
                self.param_index[op.members[0].scope + "/biases"] = ("W", w_off, (op.K,), None)
                w_off += (op.K + 7) // 8 * 8
            if op.kind in ("bn", "frozen"):
                op.beta_off = bt_off
                ch = 0
                for m in op.members:
                    self.param_index[m.scope + "/BatchNorm/beta"] = ("Bt", bt_off + ch, (m.K,), None)
                    self.param_index[m.scope + "/BatchNorm/moving_mean"] = ("MM", bt_off + ch, (m.K,), None)
                    self.param_index[m.scope + "/BatchNorm/moving_variance"] = ("MV", bt_off + ch, (m.K,), None)
                    ch += m.K
                bt_off += (op.K + 7) // 8 * 8
        self.nW, self.nBt = w_off, bt_off
        first_head = next(op for op in self.convs if op.is_head)
        self.head_w_start, self.head_bt_start = first_head.w_off, first_head.beta_off
        f32 = dict(dtype=torch.float32, device=dev)
        self.W = torch.zeros(self.nW, **f32)
        self.Wb = torch.zeros(self.nW, dtype=torch.bfloat16, device=dev)
        self.Bt = torch.zeros(self.nBt, **f32)
        self.MM = torch.zeros(self.nBt, **f32)
        self.MV = torch.ones(self.nBt, **f32)
        self.bn_mean = torch.zeros(self.nBt, **f32)
        self.bn_var = torch.zeros(self.nBt, **f32)      # biased batch variance (deferred moving-average update)
        self.bn_rstd = torch.ones(self.nBt, **f32)
        self.bn_scale = torch.ones(self.nBt, **f32)     # folded (frozen) BN
        self.bn_shift = torch.zeros(self.nBt, **f32)
        self.init_weights(seed)
        if self.mode == "train":
            # ONE gradient buffer [beta gradients | step control block | filter / bias gradients]: cleared by one fill, and
            # the LAST all-reduce bucket of a data-parallel step (the bottom of the network, Trainer.step) takes the beta
            # gradients and the control block along instead of a second small collective behind it.
            # beta gradients + the STEP CONTROL BLOCK (8 floats behind them; zeroed with the gradients, summed over ranks
            # with them): [0] grid-barrier timeouts of this step's one-launch BN backward, [1] ranks asking to stop.
            # The optimiser launches test it (skip_ctl, include/mbx.h) and leave a flagged step un-applied.
            # filter gradients on a 256-byte boundary (the pad is all-reduced along: zeros); on a 32-byte boundary the step
            # measured 1.3 % slower (3699 vs 3747 images/s, same box: the weight-gradient atomics and the optimiser pass)
            self.G_off = (self.nBt + 8 + 63) // 64 * 64
            assert self.G_off % 8 == 0
            self.G = torch.zeros(self.G_off + self.nW, **f32)
            self.Btg = self.G[:self.nBt + 8]
            self.Wg = self.G[self.G_off:]
            self.step_ctl = self.Btg[self.nBt:self.nBt + 8]
        self._alloc_scratch()
        self._build_backward()
        if torch.device(self.dev).type == "cuda":
            self.prepare_filters()
            if os.environ.get("MBX_TUNE_SAVE"):           # refresh the shipped table of measured tile choices
                ops.save_tune_cache()

    def init_weights(self, seed):
        """slim defaults (un-vendored): Xavier-uniform filters, zero biases/betas, moving mean 0 / variance 1."""
        gen = torch.Generator().manual_seed(seed)
        W = torch.zeros(self.nW, dtype=torch.float32)
        for name, (buf, off, shape, cpad) in self.param_index.items():
            if buf == "W" and name.endswith("/weights"):
                K, R, S_, Cc = shape
                lim = math.sqrt(6.0 / (R * S_ * Cc + R * S_ * K))
                n = K * R * S_ * Cc
                v = ((torch.rand(n, generator=gen) * 2 - 1) * lim).reshape(shape)
                W[off:off + K * R * S_ * cpad].reshape(K, R, S_, cpad)[..., :Cc] = v
        self.W.copy_(W)
        self.Wb.copy_(self.W.to(torch.bfloat16))

    def get_param(self, name, kind="value"):
        """View of a reference variable inside the flat buffers (kind: value | grad)."""
        buf, off, shape, cpad = self.param_index[name]
        t = {"W": self.W, "Bt": self.Bt, "MM": self.MM, "MV": self.MV}[buf] if kind == "value" else \
            {"W": self.Wg, "Bt": self.Btg}[buf]
        if cpad is not None and cpad != shape[-1]:
            K, R, S_, Cc = shape
            return t[off:off + K * R * S_ * cpad].reshape(K, R, S_, cpad)[..., :Cc]
        n = int(np.prod(shape))
        return t[off:off + n].reshape(shape)

    def set_param(self, name, value):
        self.get_param(name).copy_(torch.as_tensor(value, dtype=torch.float32).reshape(self.get_param(name).shape))

    def refresh_bf16(self):
        """Call after editing W directly (set_param, checkpoint load)."""
        self.Wb.copy_(self.W.to(torch.bfloat16))
        self.prepare_filters()

    def _alloc_scratch(self):
        dev = self.dev
        max_y = max_stats = max_bwd = n_stats16 = 0
        self.y_tmp = {}
        l = _lib.lib()
        # batch-norm groups that apply: training-mode BN on every member, same pixels / relu / trainability, consecutive in
        # forward order, outputs in ONE buffer, betas contiguous, the group small enough for the lane-owns-a-channel-group kernels
        for op in self.convs:
            op.group = None
        for members in self._bn_group_requests:
            idx = [self.fwd.index(m) for m in members]
            ok = all(m.kind == "bn" and m.M == members[0].M and m.relu == members[0].relu and m.trainable == members[0].trainable and
                     m.out.buf is members[0].out.buf and m.out.ld == members[0].out.ld and m.K % 8 == 0 for m in members)
            ok = ok and idx == list(range(idx[0], idx[0] + len(idx))) and len(members) <= 4
            ok = ok and all(b.beta_off == a.beta_off + a.K for a, b in zip(members, members[1:]))
            g = BnGroup(members)
            ok = ok and g.K <= 2048 and g.chan_map([m.out for m in members])[0] >= 0
            if ok:
                self.bn_groups.append(g)
                for m in members:
                    m.group = g
        for op in self.convs:
            if op.kind == "bn":
                g = op.group
                lead = g is None or op is g.members[0]
                K_all = op.K if g is None else g.K
                N_, H_, W_ = op.out.N, op.out.H, op.out.W
                if lead:
                    ybuf = torch.empty((op.M, K_all), dtype=torch.bfloat16, device=dev)      # pre-BN output, kept for backward
                    self._bufs.append(ybuf)
                    if g is not None:
                        g.y = ybuf
                else:
                    ybuf = g.y
                ko = 0 if g is None else g.koff[g.members.index(op)]
                op.y_view = View(ybuf, N_, H_, W_, op.K, K_all, ko)
                if lead:
                    op.stats16_base = n_stats16                 # [STATS_ROWS][K_all][2] int64 (fixed point) of the layer / group, in float units
                    n_stats16 += STATS_ROWS * K_all * 4
                op.stats16_off = (op.stats16_base if g is None else g.members[0].stats16_base) + 4 * ko
                d = self._desc(op, op.y_view)
                rows = max(ops.conv_stats_rows(d), (op.M + 63) // 64)                                   # any tile height
                if g is None and ops.splitk_slices(d, self.n_cus):
                    rows = max(rows, (op.M + 15) // 16)                                                   # split-K: a row per 16 pixels
                if g is None:
                    op.stats_off = 0
                    max_stats = max(max_stats, rows * op.K * 2)
                else:
                    i = g.members.index(op)
                    g.stats_off[i] = 0 if i == 0 else g.stats_off[i - 1] + g._bound
                    g._bound = rows * op.K * 2
                    op.stats_off = g.stats_off[i]
                    max_stats = max(max_stats, op.stats_off + g._bound)
                if op.trainable:
                    # dy of EVERY layer stays alive until the grouped weight-gradient launch at the end of its backward
                    # segment (3.2 GB at BATCH_SIZE 64: sized for 288 GB of HBM, not for reuse)
                    if lead:
                        dybuf = torch.empty((op.M, K_all), dtype=torch.bfloat16, device=dev)
                        self._bufs.append(dybuf)
                        if g is not None:
                            g.dy = dybuf
                        max_bwd = max(max_bwd, l.mbx_bn_bwd_rows(op.M, K_all) * K_all * 2,
                                      l.mbx_bn_bwd_rows_pooled(N_, H_, W_, K_all) * K_all * 2)       # (either form of the three-launch backward)
                    else:
                        dybuf = g.dy
                    op.dy_view = View(dybuf, N_, H_, W_, op.K, K_all, ko)
        # one-launch BN backward: per-layer (per-group) accumulators + arrival counter, zeroed with the gradients every step
        ws_floats = 0
        for op in self.convs:
            op.bn_ws_off = -1
            g = getattr(op, "group", None)
            if g is not None and op is not g.members[0]:
                op.bn_ws_off = g.members[0].bn_ws_off
                continue
            K_all = op.K if g is None else g.K
            if op.kind == "bn" and op.trainable and torch.device(dev).type == "cuda" and l.mbx_bn_bwd_onepass_supported(op.M, K_all, self.chain_cap or self.bn_max_wg):
                op.bn_ws_off = ws_floats
                ws_floats += (l.mbx_bn_bwd_onepass_workspace_bytes(K_all) // 4 + 7) // 8 * 8
        self._plan_bw_stats()
        for op in self.convs:
            op.bw_rows_off = -1
            u = getattr(op, "bw_unit", None)
            if u is not None:
                lead = u[0]
                if op is lead:
                    op.bw_rows_off = ws_floats                       # [STATS_ROWS][K_all][2], cleared with the workspace every step
                    ws_floats += STATS_ROWS * sum(m.K for m in u) * 2
                else:
                    op.bw_rows_off = lead.bw_rows_off
        # fused data-gradient + BN-backward launches: accumulators per batch-norm unit, a grid-barrier control block per launch
        self._plan_fused_bwd()
        for op in self.convs:
            op.fb_acc_off = op.fb_bar_off = -1
        for op in self.convs:
            u = op.fb_unit
            if u is not None and op is u[0]:
                ws_floats = (ws_floats + 31) // 32 * 32
                op.fb_acc_off = ws_floats
                ws_floats += ops.BN_BWD_SLOTS * 2 * sum(m.K for m in u)
            if op.fb_segments is not None:
                ws_floats = (ws_floats + 31) // 32 * 32
                op.fb_bar_off = ws_floats
                ws_floats += ops.GRID_BARRIER_BYTES // 4
        ws_floats = (ws_floats + 31) // 32 * 32
        self.bn_ws = torch.zeros(max(ws_floats, 8), dtype=torch.float32, device=dev)
        assert self.bn_ws.data_ptr() % 128 == 0 or torch.device(dev).type != "cuda"
        # ONE buffer cleared by one fill at the start of forward(): [work counters of the persistent launches | statistics rows]
        n_ctr = self.i5_counters.numel()
        # (+ the grid-barrier control blocks of the fused convolution + BN-apply launches: one per batch-norm convolution, 6.4 KB each)
        n_stats16 = (max(n_stats16, 4) + 31) // 32 * 32
        self._n_fwd_barriers = sum(1 for op in self.convs if op.kind == "bn") if self.fuse_apply else 0
        self._fwd_barriers_used = 0
        bar_words = ops.GRID_BARRIER_BYTES // 4
        self._fwd_clear = torch.zeros(n_ctr + n_stats16 + self._n_fwd_barriers * bar_words + 32, dtype=torch.int32, device=dev)
        self.i5_counters = self._fwd_clear[:n_ctr]
        self.stats16 = self._fwd_clear[n_ctr:n_ctr + n_stats16].view(torch.float32)
        b0 = n_ctr + n_stats16
        b0 += (-(self._fwd_clear.data_ptr() // 4 + b0)) % 32                      # 128-byte aligned control blocks
        self._fwd_barrier_base = self._fwd_clear.data_ptr() + 4 * b0
        self.bn_thr = torch.zeros(max(self.nBt, 8), dtype=torch.float32, device=dev)    # relu threshold on y per channel (mbx_bn_apply_fused_mapped)
        self.bn_timeouts_total = torch.zeros((), dtype=torch.int64, device=dev)    # workgroups that gave up on a grid barrier, ever
        self.stats_scratch = torch.zeros(max(max_stats, 2), dtype=torch.float32, device=dev)
        self.bwd_scratch = torch.zeros(max(max_bwd, 2), dtype=torch.float32, device=dev)
        self.m12 = torch.zeros(2 * 2048, dtype=torch.float32, device=dev)
        self.reg_loss = torch.zeros(1, dtype=torch.float32, device=dev)

    def _plan_bw_stats(self):
        """Which batch-norm layers (groups) get their backward statistics from the data gradient(s) that write their
        activation gradient, and which data gradients carry the statistics epilogue.  A consumer convolution X is CONVERTIBLE
        when it is a trainable stride-1 convolution with a data gradient whose input view is tiled exactly by outputs of
        eligible batch-norm members (segments at multiples of 32 channels, at most 4) and nothing else reads those channels;
        a unit is ELIGIBLE when every channel of every member has exactly one consumer and that one is convertible.  Fixed point
        of the two.  Sets op.bw_unit (tuple of the unit's members) on eligible layers and op.bw_segments on convertible consumers:
        [(first channel relative to X's input view, member, first channel inside the member's output, channels)]."""
        for op in self.convs:
            op.bw_unit, op.bw_segments = None, None
        if not self.bw_stats or torch.device(self.dev).type != "cuda":
            return
        units = []
        for op in self.convs:
            if op.kind != "bn" or not op.trainable:
                continue
            g = op.group
            if g is not None and op is not g.members[0]:
                continue
            mem = tuple([op] if g is None else g.members)
            if sum(m.K for m in mem) <= 2048:
                units.append(mem)
        unit_of = {id(m): u for u in units for m in u}

        def member_at(buf, c):
            for u in units:
                for m in u:
                    if m.out.buf is buf and m.out.ch_off <= c < m.out.ch_off + m.K:
                        return m
            return None

        def readers(buf, lo, hi):
            """forward ops that read channels [lo, hi) of buf: as their input, or as a residual block's skip (trunk) tensor"""
            r = [o for o in self.fwd if o.x.buf is buf and o.x.ch_off < hi and o.x.ch_off + o.x.C > lo]
            r += [o for o in self.fwd if isinstance(o, ConvOp) and o.skip is not None and o.skip.buf is buf and
                  o.skip.ch_off < hi and o.skip.ch_off + o.skip.C > lo]
            return r

        def segments(X):
            """tiling of X's input view by batch-norm member outputs, or None"""
            segs, c, end = [], X.x.ch_off, X.x.ch_off + X.x.C
            while c < end:
                m = member_at(X.x.buf, c)
                if m is None:
                    return None
                n = min(end, m.out.ch_off + m.K) - c
                segs.append((c - X.x.ch_off, m, c - m.out.ch_off, n))
                c += n
            if len(segs) > 4 or any(sg[0] % 32 for sg in segs):
                return None
            return segs

        eligible = set(id(u) for u in units)
        while True:
            conv_ok = {}
            for X in self.fwd:
                if not isinstance(X, ConvOp) or not X.need_dx or not X.trainable or X.stride != 1:
                    continue
                segs = segments(X)
                if segs is None or any(id(unit_of[id(sg[1])]) not in eligible for sg in segs):
                    continue
                if len(readers(X.x.buf, X.x.ch_off, X.x.ch_off + X.x.C)) != 1:
                    continue                                   # someone else reads (part of) these channels: accumulated gradient
                # (the direct 3x3 launch has no statistics epilogue: the stem's layers stay on the grid-barrier form)
                dd = ops.ConvDesc()                               # (the fields ops.direct3_applies looks at, of X's data gradient)
                dd.R, dd.S, dd.stride, dd.epilogue, dd.C_in, dd.C_out = X.R, X.S, 1, ops.EPI_STORE, X.K, X.Cin
                dd.N, dd.H_out, dd.W_out, dd.rscale = X.x.N, X.x.H, X.x.W, (X.rscale if X.kind == "residual" else 0.0)
                dd.W_out = X.x.W
                dd.H_in, dd.W_in, dd.H_out = X.x.H, X.x.W, X.x.H
                if ops.direct3_applies(dd) or ops.directw_applies(dd) or (X.out.H == X.x.H and X.out.W == X.x.W and ops.resident_applies(dd)):
                    continue
                conv_ok[id(X)] = segs
            drop = set()
            for u in units:
                if id(u) not in eligible:
                    continue
                for m in u:
                    rd = readers(m.out.buf, m.out.ch_off, m.out.ch_off + m.K)
                    covered = sorted((max(o.x.ch_off, m.out.ch_off), min(o.x.ch_off + o.x.C, m.out.ch_off + m.K)) for o in rd)
                    tiled = bool(covered) and covered[0][0] == m.out.ch_off and covered[-1][1] == m.out.ch_off + m.K and \
                        all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
                    if not tiled or any(id(o) not in conv_ok for o in rd):
                        drop.add(id(u))
            if not drop:
                break
            eligible -= drop
        for u in units:
            if id(u) in eligible:
                for m in u:
                    m.bw_unit = u
        for X in self.fwd:
            if isinstance(X, ConvOp) and id(X) in conv_ok:
                X.bw_segments = conv_ok[id(X)]

    def _plan_fused_bwd(self):
        """Which batch-norm layers (groups) have their backward run by the data gradient(s) that write their activation
        gradient (mbx_conv_desc.bn_bwd), and which data gradients carry that tail.  A consumer convolution X is CONVERTIBLE
        when it is a trainable stride-1 convolution with a data gradient whose input view is tiled exactly by outputs of
        eligible batch-norm members (segments at multiples of 8 channels, at most 4), nothing else reads those channels,
        and its data gradient is not one of the stem's direct launches; a unit is ELIGIBLE when every channel of every
        member has exactly one consumer and that one is convertible.  Fixed point of the two.  Sets op.fb_unit (tuple of the
        unit's members) on eligible layers and op.fb_segments on convertible consumers:
        [(first channel relative to X's input view, member, first channel inside the member's output, channels)]."""
        for op in self.convs:
            op.fb_unit, op.fb_segments = None, None
        if not self.fuse_bwd:
            return
        units = []
        for op in self.convs:
            if op.kind != "bn" or not op.trainable:
                continue
            g = op.group
            if g is not None and op is not g.members[0]:
                continue
            mem = tuple([op] if g is None else g.members)
            if sum(m.K for m in mem) <= 2048 and all(getattr(m, "fused_pool", None) is None for m in mem):
                units.append(mem)
        unit_of = {id(m): u for u in units for m in u}
        pairs_ok = os.environ.get("MBX_FUSE_BWD_PAIRS", "1") != "0"
        only = re.compile(os.environ["MBX_FUSE_BWD_ONLY"]) if os.environ.get("MBX_FUSE_BWD_ONLY") else None   # consumers by name (A/B)

        def member_at(buf, c):
            for u in units:
                for m in u:
                    if m.out.buf is buf and m.out.ch_off <= c < m.out.ch_off + m.K:
                        return m
            return None

        def readers(buf, lo, hi):
            r = [o for o in self.fwd if o.x.buf is buf and o.x.ch_off < hi and o.x.ch_off + o.x.C > lo]
            r += [o for o in self.fwd if isinstance(o, ConvOp) and o.skip is not None and o.skip.buf is buf and
                  o.skip.ch_off < hi and o.skip.ch_off + o.skip.C > lo]
            return r

        def segments(X):
            segs, c, end = [], X.x.ch_off, X.x.ch_off + X.x.C
            while c < end:
                m = member_at(X.x.buf, c)
                if m is None:
                    return None
                n = min(end, m.out.ch_off + m.K) - c
                segs.append((c - X.x.ch_off, m, c - m.out.ch_off, n))
                c += n
            if len(segs) > 4 or any(sg[0] % 8 for sg in segs):
                return None
            return segs

        eligible = set(id(u) for u in units)
        conv_ok = {}
        while True:
            conv_ok = {}
            for X in self.fwd:
                if not isinstance(X, ConvOp) or not X.need_dx or not X.trainable or X.stride != 1:
                    continue
                if X.x.img_stride != X.x.H * X.x.W * X.x.ld or X.Cin % 8 or (only is not None and not only.search(X.name)):
                    continue
                if not pairs_ok and X.kind == "bn" and X.group is not None and len(X.group.members) == 2:
                    continue
                segs = segments(X)
                if segs is None or any(id(unit_of[id(sg[1])]) not in eligible for sg in segs):
                    continue
                if len(readers(X.x.buf, X.x.ch_off, X.x.ch_off + X.x.C)) != 1:
                    continue                                   # someone else reads (part of) these channels: accumulated gradient
                dd = ops.ConvDesc()                               # (the fields ops.direct3_applies looks at, of X's data gradient)
                dd.R, dd.S, dd.stride, dd.epilogue, dd.C_in, dd.C_out = X.R, X.S, 1, ops.EPI_STORE, X.K, X.Cin
                dd.N, dd.H_in, dd.W_in, dd.H_out, dd.W_out = X.x.N, X.out.H, X.out.W, X.x.H, X.x.W
                if ops.direct3_applies(dd):
                    continue                                   # (the stem: the direct launch has no tail; its layers keep their own backward)
                # a tail walks its workgroup's tiles one after the other: on one tile it costs what the BN-backward launch costs, on
                # several it costs more (block35 at BATCH_SIZE 64: 5 tails of 15-40 us for 3 launches of 17 us, +0.37 ms per step)
                if ops.fused_tail_rounds(dd, self.n_cus) > int(os.environ.get("MBX_FUSE_BWD_MAX_ROUNDS", "1")):
                    continue
                conv_ok[id(X)] = segs
            drop = set()
            for u in units:
                if id(u) not in eligible:
                    continue
                for m in u:
                    rd = readers(m.out.buf, m.out.ch_off, m.out.ch_off + m.K)
                    covered = sorted((max(o.x.ch_off, m.out.ch_off), min(o.x.ch_off + o.x.C, m.out.ch_off + m.K)) for o in rd)
                    tiled = bool(covered) and covered[0][0] == m.out.ch_off and covered[-1][1] == m.out.ch_off + m.K and \
                        all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
                    if not tiled or any(id(o) not in conv_ok for o in rd):
                        drop.add(id(u))
            if not drop:
                break
            eligible -= drop
        for u in units:
            if id(u) in eligible:
                for m in u:
                    m.fb_unit = u
        for X in self.fwd:
            if isinstance(X, ConvOp) and id(X) in conv_ok:
                X.fb_segments = conv_ok[id(X)]

    def _fb_table(self, X):
        """mbx_bn_bwd_fused of consumer X's data gradient (kept alive in self._keep_fb)."""
        t = ops.BnBwdFused()
        t.barrier = self.bn_ws.data_ptr() + 4 * X.fb_bar_off
        t.n = len(X.fb_segments)
        t.step_poison = self.step_ctl.data_ptr()
        for i, (c_rel, m, c_in, n) in enumerate(X.fb_segments):
            lead = m.fb_unit[0]
            K_all = sum(q.K for q in m.fb_unit)
            ko = m.y_view.ch_off                                         # the member's first channel inside the unit's [M, K_all] tensors
            assert m.y_view.ld == K_all and m.dy_view.ld == K_all and lead.fb_acc_off >= 0
            t.c_begin[i] = c_rel
            t.y[i], t.ld_y[i] = m.y_view.buf.data_ptr() + 2 * (ko + c_in), K_all
            t.dy[i], t.ld_dy[i] = m.dy_view.buf.data_ptr() + 2 * (ko + c_in), K_all
            t.mean[i] = self.bn_mean.data_ptr() + 4 * (m.beta_off + c_in)
            t.rstd[i] = self.bn_rstd.data_ptr() + 4 * (m.beta_off + c_in)
            t.beta[i] = self.Bt.data_ptr() + 4 * (m.beta_off + c_in)
            t.dbeta[i] = self.Btg.data_ptr() + 4 * (m.beta_off + c_in)
            t.acc[i], t.acc_ld[i] = self.bn_ws.data_ptr() + 4 * (lead.fb_acc_off + ko + c_in), K_all
            t.relu[i] = int(m.relu)
        self._keep_fb = getattr(self, "_keep_fb", []) + [t]
        return t

    def _bw_table(self, X):
        """mbx_bn_bwd_stats of consumer X's data gradient (kept alive in self._keep_bw)."""
        t = _lib.BnBwdStats()
        t.n, t.rows_mod = len(X.bw_segments), STATS_ROWS
        for i, (c_rel, m, c_in, n) in enumerate(X.bw_segments):
            lead = m.bw_unit[0]
            K_all = sum(q.K for q in m.bw_unit)
            ko = m.y_view.ch_off                                         # the member's first channel inside the unit's [M, K_all] tensors
            t.c_begin[i] = c_rel
            t.y[i], t.ld_y[i] = m.y_view.buf.data_ptr() + 2 * (ko + c_in), K_all
            t.relu_thr[i] = self.bn_thr.data_ptr() + 4 * (m.beta_off + c_in)
            t.stats[i], t.stats_ld[i] = self.bn_ws.data_ptr() + 4 * (lead.bw_rows_off + 2 * (ko + c_in)), K_all
        self._keep_bw = getattr(self, "_keep_bw", []) + [t]
        return t

    # ------------------------------------------------------------------ descriptors
    def _w(self, op):
        return self.Wb[op.w_off:]

    def _desc(self, op, y: View, **kw):
        return ops.make_desc(op.x, self._w(op), op.K, op.R, op.S, op.stride, op.pad_t, op.pad_l, y, **kw)

    def _sl(self, t, off, n):
        return t[off:off + n]

    def _tune(self, op, d, what):
        """Pick the tile configuration of one conv launch by measurement (ops.autotune); a no-op on CPU."""
        key = (what, op.x.N, op.x.H, op.x.W, op.x.C, op.K, op.R, op.S, op.stride, op.pad_t, op.pad_l, d.C_out, d.C_in,
               d.epilogue, bool(d.stats_partial), d.accumulate, bool(d.skip) or (bool(d.relu_bits) and d.epilogue == ops.EPI_STORE), d.relu)
        if d.relu_bits and d.epilogue == ops.EPI_RESIDUAL:
            key = key + ("bits",)      # the training forward of a residual block also writes the sign bits: its own table entry
        self.tune_registry.append((repr(key), d, what))           # tools/tune_in_situ.py re-measures these inside a step
        if self.autotune:
            ops.autotune(d, key)
            # a MEASURED exception to the launch rules below (tools/tune_by_trace.py writes one where a table tile beat the
            # rule's launch inside the replayed step: e.g. the resident-image launch on block8's 8 x 8 maps at BATCH_SIZE 256,
            # 1024 one-shot tiles in four rounds, each paying its operand landing)
            over = ops._TUNED.get(repr(key) + "#norule") if os.environ.get("MBX_NO_RULE_EXCEPTIONS") != "1" else None
            if over is not None:
                keep = d.tile_config
                d.tile_config = int(over)
                if _lib.lib().mbx_conv_supported(C.byref(d)) != 0:
                    d.tile_config, over = keep, None
            if os.environ.get("MBX_NO_I5") == "1" and d.tile_config > ops.I5_FLAG:   # A/B knob
                d.tile_config = ops._TUNED.get(repr(key) + "#i3", 0)       # the best igemm3 tile the tuner saw, else the rule
            # The persistent igemm5 launch needs a whole CU per workgroup; in data-parallel runs RCCL's kernels hold CUs
            # for milliseconds, and a workgroup that starts late would finish a STATIC share of the tiles late.  Every
            # igemm5 launch therefore gets a work counter (tiles after a workgroup's first are pulled from it, like the
            # items of the grouped weight gradient) -- the same kernels with and without torch.distributed.
            if d.tile_config > ops.I5_FLAG and os.environ.get("MBX_I5_STATIC") != "1":
                n = ops.I7_COUNTERS if d.tile_config == ops.I7_TILE_CONFIG else 1        # igemm7: one counter per column tile
                assert self._i5_used + n <= self.i5_counters.numel()
                d.work_counter = self.i5_counters.data_ptr() + 4 * self._i5_used
                self._i5_used += n
            if os.environ.get("MBX_NO_2STAGE") == "1":                     # bisecting aid: 3-deep-ring twins of the 2-deep tiles
                d.tile_config = {9: 7, 10: 2, 11: 5, 12: 2, 13: 8, 14: 1}.get(d.tile_config, d.tile_config)
            # few channels on a large map (the stem's 3x3 layers, forward and data gradient): the direct launch stages each
            # pixel patch once instead of gathering it nine times (by rule, not by the table; bit-identical outputs)
            if over is not None:
                pass
            elif ops.direct3_applies(d):
                d.tile_config, d.work_counter = ops.DIRECT3_TILE_CONFIG, None
                _lib.check(_lib.lib().mbx_conv_supported(C.byref(d)), "direct 3x3 " + op.name)
            elif ops.directw_applies(d) and not (what == "fwd" and d.stats_partial and getattr(op, "group", None) is not None
                                                 and len(op.group.members) == 2 and os.environ.get("MBX_CONV_PAIR", "1") != "0"):
                # few channels on a NARROW map (block35's 3x3 layers, 35 x 35): the whole-width direct launch -- except for the
                # two sibling FORWARD convolutions of a batch-norm group in training, which the implicit GEMM runs as ONE pair
                # launch (21.7 us against 12.0 + 11.3 as two direct launches: these launches are latency-bound)
                d.tile_config, d.work_counter = ops.DIRECTW_TILE_CONFIG, None
                _lib.check(_lib.lib().mbx_conv_supported(C.byref(d)), "direct 3x3 (whole-width) " + op.name)
            elif ops.pwres_applies(d):
                # the 1x1 launches whose time is their epilogue's trunk traffic (residual "up" forward, accumulate + mask data
                # gradients): pixels resident, filter streamed, one small epilogue per 128 output channels (by rule; bit-identical)
                d.tile_config, d.work_counter = ops.PWRES_TILE_CONFIG, None
                _lib.check(_lib.lib().mbx_conv_supported(C.byref(d)), "pixel-resident 1x1 " + op.name)
            elif ops.resident_applies(d):
                # many channels on a SMALL map with a multi-tap filter (block17's 1x7 / 7x1 layers, forward and data gradient): the
                # resident-image launch stages each image once instead of gathering it once per tap (by rule; bit-identical outputs)
                d.tile_config, d.work_counter = ops.RESIDENT_TILE_CONFIG, None
                _lib.check(_lib.lib().mbx_conv_supported(C.byref(d)), "resident image " + op.name)
        return d

    def _build_forward_launches(self):
        """Pre-bind every forward launch; returns a list of zero-argument callables."""
        L = []
        l = _lib.lib()
        st = lambda: torch.cuda.current_stream().cuda_stream
        for op in self.fwd:
            if isinstance(op, PoolOp):
                if not getattr(op, "fused_fwd", False):          # (else: its producer's BN apply writes the pooled tensor)
                    L.append(op.forward)
                continue
            if op.kind == "bn" and op.group is not None:
                # member of a batch-norm group: its convolution writes its channel slice of the group's [M, K] tensor and its
                # own statistics partials; the LAST member's launch is followed by ONE finalize and ONE apply for the group
                g = op.group
                i = g.members.index(op)
                use16 = self.atomic_stats
                if use16:
                    d = self._tune(op, self._desc(op, op.y_view, stats=self.stats16[op.stats16_off:], stats_rows_mod=STATS_ROWS,
                                                  stats_ld=g.K), "fwd")
                else:
                    d = self._tune(op, self._desc(op, op.y_view, stats=self.stats_scratch[op.stats_off:]), "fwd")
                g.rows[i] = ops.conv_stats_rows(d)
                assert use16 or g.rows[i] * op.K * 2 <= (g.stats_off[i + 1] - g.stats_off[i] if i + 1 < len(g.members) else 1 << 62)
                g.fwd_desc[i] = d
                if i + 1 < len(g.members):
                    # (a group of two whose convolutions resolve to the same small-tile kernel run as ONE launch, issued with the
                    # last member: mbx_conv_pair; decided below, when both descriptors exist)
                    L.append(lambda d=d, op=op, g=g: None if g.pair_fwd else _lib.check(l.mbx_conv(C.byref(d), st()), op.name))
                    continue
                if len(g.members) == 2 and torch.device(self.dev).type == "cuda" and os.environ.get("MBX_CONV_PAIR", "1") != "0":
                    g.pair_fwd = l.mbx_conv_pair(C.byref(g.fwd_desc[0]), C.byref(d), st()) == 0        # (a real launch, like the tuner's)
                lead, n = g.members[0], len(g.members)
                parts = (C.c_void_p * n)(*[self.stats_scratch.data_ptr() + 4 * o for o in g.stats_off])
                rows_a, cs_a = (C.c_int32 * n)(*g.rows), (C.c_int32 * n)(*[m.K for m in g.members])
                base, amap = g.chan_map([m.out for m in g.members])
                a_ptr = op.out.buf.data_ptr() + 2 * base
                mean, rstd = self._sl(self.bn_mean, lead.beta_off, g.K), self._sl(self.bn_rstd, lead.beta_off, g.K)
                mm, mv = self._sl(self.MM, lead.beta_off, g.K), self._sl(self.MV, lead.beta_off, g.K)
                var, beta = self._sl(self.bn_var, lead.beta_off, g.K), self._sl(self.Bt, lead.beta_off, g.K)
                thr = self._sl(self.bn_thr, lead.beta_off, g.K)
                s16 = self.stats16[lead.stats16_base:]

                def run(d=d, op=op, g=g, parts=parts, rows_a=rows_a, cs_a=cs_a, amap=amap, a_ptr=a_ptr, mean=mean, rstd=rstd,
                        mm=mm, mv=mv, var=var, beta=beta, n=n, use16=use16, thr=thr, s16=s16):
                    s = st()
                    if g.pair_fwd:
                        _lib.check(l.mbx_conv_pair(C.byref(g.fwd_desc[0]), C.byref(d), s), "pair " + op.name)
                    else:
                        _lib.check(l.mbx_conv(C.byref(d), s), op.name)
                    decay = self.bn_decay
                    if self.defer_moving:
                        mm, mv, decay = None, var, -1.0
                    if use16:
                        # every member ADDED its tile sums into the group's 16 rows: the apply launch reduces them itself
                        _lib.check(l.mbx_bn_apply_fused_mapped(s16.data_ptr(), STATS_ROWS, op.M, BN_EPS, decay, g.y.data_ptr(), op.M,
                                                               g.K, beta.data_ptr(), int(op.relu), a_ptr, op.out.ld, C.byref(amap),
                                                               mean.data_ptr(), rstd.data_ptr(), ops._p(mm), ops._p(mv),
                                                               thr.data_ptr(), s), "bn_apply_fused_mapped")
                        return
                    _lib.check(l.mbx_bn_finalize_parts(parts, rows_a, cs_a, n, op.M, BN_EPS, decay, mean.data_ptr(), rstd.data_ptr(),
                                                       ops._p(mm), ops._p(mv), s), "bn_finalize_parts")
                    _lib.check(l.mbx_bn_apply_mapped(g.y.data_ptr(), op.M, g.K, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(),
                                                     int(op.relu), a_ptr, op.out.ld, C.byref(amap), s), "bn_apply_mapped")
                L.append(run)
            elif op.kind == "bn":
                yv = op.y_view
                fpool = getattr(op, "fused_pool", None)
                use16 = self.atomic_stats and fpool is None          # (the fused apply + max-pool launch reads finalized statistics)
                if use16:
                    d = self._tune(op, self._desc(op, yv, stats=self.stats16[op.stats16_off:], stats_rows_mod=STATS_ROWS), "fwd")
                else:
                    d = self._tune(op, self._desc(op, yv, stats=self.stats_scratch), "fwd")
                S_ = ops.splitk_slices(d, self.n_cus) if torch.device(self.dev).type == "cuda" else 0
                if S_:
                    # long K, few tiles (the 3x3 head convolutions on the 1536-channel map): K slices + a reduce launch
                    d.tile_config = ops.SPLITK_FLAG + S_
                    # (a workspace per launch: the head scales run on their own streams, Net.forward)
                    need = int(l.mbx_conv_splitk_workspace_bytes(C.byref(d)))
                    ws = torch.empty(need // 4, dtype=torch.float32, device=self.dev)
                    self._splitk_ws = getattr(self, "_splitk_ws", []) + [ws]
                    d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
                    _lib.check(l.mbx_conv_supported(C.byref(d)), "split-K " + op.name)
                rows = ops.conv_stats_rows(d)
                mean, rstd = self._sl(self.bn_mean, op.beta_off, op.K), self._sl(self.bn_rstd, op.beta_off, op.K)
                mm, mv = self._sl(self.MM, op.beta_off, op.K), self._sl(self.MV, op.beta_off, op.K)
                beta = self._sl(self.Bt, op.beta_off, op.K)
                out = op.out

                var = self._sl(self.bn_var, op.beta_off, op.K)
                thr = self._sl(self.bn_thr, op.beta_off, op.K)
                s16 = self.stats16[op.stats16_off:]

                if fpool is not None:
                    fpool.fused_fwd = True
                # the layer's BN apply as the tail of the convolution launch (mbx_conv_desc.bn_apply), where the library has it
                fd = ba = None
                if use16 and self.fuse_apply and not S_ and torch.device(self.dev).type == "cuda" and out.img_stride == out.H * out.W * out.ld \
                        and (not os.environ.get("MBX_FUSE_APPLY_ONLY") or re.search(os.environ["MBX_FUSE_APPLY_ONLY"], op.name)):
                    ba = ops.BnApplyDesc()
                    ba.barrier = self._fwd_barrier_base + ops.GRID_BARRIER_BYTES * self._fwd_barriers_used
                    ba.a, ba.ld_a, ba.beta, ba.mean, ba.rstd = out.ptr, out.ld, beta.data_ptr(), mean.data_ptr(), rstd.data_ptr()
                    ba.relu_thr, ba.relu, ba.eps, ba.step_poison = thr.data_ptr(), int(op.relu), BN_EPS, self.step_ctl.data_ptr()
                    fd = ops.ConvDesc.from_buffer_copy(d)
                    fd.bn_apply = C.addressof(ba)
                    fd.work_counter = None                   # (static deal: fused_bn.h)
                    if l.mbx_conv_supported(C.byref(fd)) == 0:
                        assert self._fwd_barriers_used < self._n_fwd_barriers
                        self._fwd_barriers_used += 1
                        self.fused_apply_launches += 1
                        self._keep_ba = getattr(self, "_keep_ba", []) + [ba]
                    else:
                        fd = ba = None

                def run(d=d, rows=rows, op=op, mean=mean, rstd=rstd, mm=mm, mv=mv, beta=beta, out=out, var=var, fpool=fpool,
                        use16=use16, thr=thr, s16=s16, fd=fd, ba=ba):
                    s = st()
                    decay = self.bn_decay
                    if self.defer_moving:                    # store mode: batch variance -> bn_var, moving statistics untouched
                        mm, mv, decay = None, var, -1.0
                    if fd is not None and not self.no_onepass:
                        # ONE launch: convolution, grid barrier, statistics, normalise + beta + relu of the tiles each workgroup wrote
                        ba.moving_mean, ba.moving_var, ba.decay = ops._p(mm), ops._p(mv), decay
                        fd.max_workgroups = self.cu_cap or self.bn_max_wg
                        _lib.check(l.mbx_conv(C.byref(fd), s), op.name + " + bn apply")
                        return
                    _lib.check(l.mbx_conv(C.byref(d), s), op.name)
                    if use16:
                        _lib.check(l.mbx_bn_apply_fused_mapped(s16.data_ptr(), STATS_ROWS, op.M, BN_EPS, decay, op.y_view.ptr, op.M,
                                                               op.K, beta.data_ptr(), int(op.relu), out.ptr, out.ld, None,
                                                               mean.data_ptr(), rstd.data_ptr(), ops._p(mm), ops._p(mv),
                                                               thr.data_ptr(), s), "bn_apply_fused_mapped")
                        return
                    if fpool is not None:
                        # the activation feeds only a 3x3 / 2 max-pool: normalise and pool in one pass, never store it
                        po = fpool.out
                        _lib.check(l.mbx_bn_finalize(self.stats_scratch.data_ptr(), rows, op.K, op.M, BN_EPS, decay,
                                                     mean.data_ptr(), rstd.data_ptr(), ops._p(mm), ops._p(mv), s), "bn_finalize")
                        _lib.check(l.mbx_bn_apply_maxpool(op.y_view.ptr, out.N, out.H, out.W, op.K, mean.data_ptr(), rstd.data_ptr(),
                                                          beta.data_ptr(), int(op.relu), po.ptr, po.img_stride, po.ld, po.H, po.W,
                                                          fpool.argmax.data_ptr(), s), "bn_apply_maxpool")
                        return
                    # (timing probe of tools/whatif_probe.py: finalize only, `a` keeps the previous step's values)
                    if self._probe_skip_apply and ("/block" in op.name or "/Block8" in op.name):
                        _lib.check(l.mbx_bn_finalize(self.stats_scratch.data_ptr(), rows, op.K, op.M, BN_EPS, decay,
                                                     mean.data_ptr(), rstd.data_ptr(), ops._p(mm), ops._p(mv), s), "fin")
                        return
                    _lib.check(l.mbx_bn_apply_fused(self.stats_scratch.data_ptr(), rows, op.M, BN_EPS, decay,
                                                    op.y_view.ptr, op.M, op.K, beta.data_ptr(), int(op.relu), out.ptr, out.ld,
                                                    mean.data_ptr(), rstd.data_ptr(), ops._p(mm), ops._p(mv), s), "bn_apply_fused")
                L.append(run)
            elif op.kind == "frozen":
                d = self._tune(op, self._desc(op, op.out, epilogue=ops.EPI_AFFINE, relu=op.relu,
                                              scale=self._sl(self.bn_scale, op.beta_off, op.K),
                                              shift=self._sl(self.bn_shift, op.beta_off, op.K)), "fwd")
                L.append(lambda d=d, op=op: _lib.check(l.mbx_conv(C.byref(d), st()), op.name))
            elif op.kind == "residual":
                d = self._tune(op, self._desc(op, op.out, epilogue=ops.EPI_RESIDUAL, relu=op.relu,
                                              shift=self._sl(self.W, op.b_off, op.K), skip=op.skip, rscale=op.rscale,
                                              relu_bits=getattr(op, "relu_bits", None)), "fwd")
                L.append(lambda d=d, op=op: _lib.check(l.mbx_conv(C.byref(d), st()), op.name))
            elif op.kind == "head":
                d = self._desc(op, op.out, epilogue=ops.EPI_STORE_F32)
                L.append(lambda d=d, op=op: _lib.check(l.mbx_conv(C.byref(d), st()), op.name))
                if op is self.heads[-1]:
                    # every head's [M_g, head_ld] float32 output -> locations / logits in prior order (model.py:296-319): ONE launch
                    L.append(lambda: _lib.check(l.mbx_head_gather_all(self.head_table, len(self.heads), self.B, self.P,
                                                                      self.locs.data_ptr(), self.logits.data_ptr(), st()), "head_gather_all"))
        return L

    # ------------------------------------------------------------------- backward
    def _gview(self, v: View) -> View:
        """Gradient view for an activation view.  Every trunk tensor of a residual stage has its OWN gradient buffer
        (G[i-1] = G[i] + dgrad is built out of place, acc_src below), so that G[i] is still intact when the deferred
        weight gradient of block i's "up" convolution reads it at the end of the segment."""
        return self.grad_of(v)

    def _claim(self, g: View):
        """Build-time bookkeeping for a write into gradient region g: returns (accumulate flag, acc_src view).  The
        first writer of a residual block's INPUT gradient adds the block's output gradient (identity path of
        model.py:21) by reading it from that other buffer."""
        key = (id(g.buf), g.ch_off, g.C)
        if key in self.written:
            return 1, None
        self.written.add(key)
        src = self.pending_acc.pop(key, None)
        return (1, src) if src is not None else (0, None)

    def _make_head_table(self):
        """mbx_head table of the detection heads (HOST array, read at launch time)."""
        arr = (_lib.Head * len(self.heads))()
        for j, op in enumerate(self.heads):
            cells, kk, off = op.head
            op.head_grad = None
            if self.mode == "train":
                op.head_grad = View(torch.zeros((op.M, self.head_ld), dtype=torch.bfloat16, device=self.dev),
                                    op.out.N, op.out.H, op.out.W, self.head_ld)
                self._bufs.append(op.head_grad.buf)
            arr[j].h, arr[j].ld_h = op.out.buf.data_ptr(), self.head_ld
            arr[j].g, arr[j].ld_g = (op.head_grad.buf.data_ptr() if op.head_grad is not None else None), self.head_ld
            arr[j].cells, arr[j].k, arr[j].off = cells, kk, off
        self.head_table = arr

    def _build_backward(self):
        self._make_head_table()
        if self.mode != "train":
            self.fwd_launches = self._build_forward_launches()
            self.bwd_launches = []
            self.bwd_jobs = []
            return
        l = _lib.lib()
        st = lambda: torch.cuda.current_stream().cuda_stream
        # dgrad filter copies ([C][R][S][Kpad], flipped): table for mbx_filter_prepare
        entries, d_off, blocks = [], 0, 0
        for op in self.convs:
            if op.need_dx:
                kpad = (op.K + 7) // 8 * 8
                e = _lib.FilterEntry(op.w_off, d_off, op.K, op.R, op.S, op.Cin, kpad, blocks)
                op.dgrad_off, op.kpad = d_off, kpad
                n = op.Cin * op.R * op.S * kpad
                d_off += (n + 7) // 8 * 8
                blocks += op.R * op.S * ((op.Cin + 31) // 32) * ((kpad + 63) // 64)
                entries.append(e)
        self.Wd = torch.zeros(max(d_off, 8), dtype=torch.bfloat16, device=self.dev)
        arr = (_lib.FilterEntry * len(entries))(*entries)
        raw = np.frombuffer(C.string_at(C.addressof(arr), C.sizeof(arr)), dtype=np.uint8).copy()
        self.filter_table = torch.from_numpy(raw).to(self.dev)
        self.filter_blocks, self.filter_entries = blocks, len(entries)

        # relu backward of a residual block output N_i = relu(...) is fused into the epilogue of the LAST
        # writer of its gradient, i.e. the data-gradient conv of the FIRST forward consumer of N_i.
        mask_of = {}               # id(consumer ConvOp) -> residual op whose output it masks
        fused_mask = set()         # id(residual op) whose relu_mask launch is not needed
        for r in self.convs:
            if r.kind == "residual" and r.relu and r.trainable:
                for c in self.fwd:
                    if isinstance(c, ConvOp) and c.x.buf is r.out.buf:
                        if c.need_dx and c.x.ch_off == r.out.ch_off and c.x.C == r.out.C:
                            mask_of[id(c)] = r
                            fused_mask.add(id(r))
                        break
                    if isinstance(c, PoolOp) and c.x.buf is r.out.buf:
                        break
        self.pending_acc = {}      # gradient region of a residual block's input -> gradient view of its output
        L = []
        self.bwd_ops = []          # the forward op each backward launch belongs to (same order as L)
        self.bwd_jobs = []         # the deferred weight-gradient job of each backward launch (None for pools)
        for op in reversed(self.fwd):
            if isinstance(op, PoolOp):
                if op.needs_backward():
                    prod = self._pool_producer(op)
                    if prod is not None:
                        # the pool's input is the activation of a batch-norm layer on the three-launch backward and nothing else
                        # reads it: that layer's backward gathers its gradient from the pool's output gradient on the fly
                        # (mbx_bn_bwd_*_pooled) -- no max-pool backward launch, its result never stored
                        prod.fused_pool = op
                        continue
                    gx = self._gview(op.x)
                    acc, src = self._claim(gx)
                    assert src is None, "a pooling layer cannot be the first writer of a residual block's input gradient"
                    L.append(op.make_backward(self._gview(op.out), gx, acc))
                    self.bwd_ops.append(op)
                    self.bwd_jobs.append(None)
                continue
            if not op.trainable:
                continue
            K, M = op.K, op.M
            dw = self._sl(self.Wg, op.w_off, K * op.R * op.S * op.Cin)
            if op.kind == "head":
                cells, kk, off = op.head
                g = op.head_grad                    # bf16 [M_g, head_ld], filled by the one scatter launch for all heads
                dyv, scale, db = g, 1.0, None
                pre = None
                if op is self.heads[-1]:            # (first in backward order)
                    pre = lambda s: _lib.check(l.mbx_head_scatter_all(self.d_locs.data_ptr(), self.d_logits.data_ptr(), self.head_table,
                                                                      len(self.heads), self.B, self.P, s), "head_scatter_all")
            elif op.kind == "residual":
                gout = self._gview(op.out)          # G[i]: complete (and relu-masked) when this op's backward runs
                gs = self._gview(op.skip)           # G[i-1] lives in its own buffer: first writer adds G[i] (acc_src)
                self.pending_acc[(id(gs.buf), gs.ch_off, gs.C)] = gout
                zb = id(self.grad_of(op.x).buf)     # per-stage branch gradient buffer is reused by every block
                self.written = {k for k in self.written if k[0] != zb}
                dyv, scale, db = gout, op.rscale, self._sl(self.Wg, op.b_off, K)
                if op.relu and id(op) not in fused_mask:
                    pre = lambda s, gout=gout, op=op: _lib.check(
                        l.mbx_relu_mask(gout.ptr, gout.ld, op.out.ptr, op.out.ld, op.M, op.K, s), "relu_mask")
                else:
                    pre = None
            else:   # bn
                dyv = op.dy_view
                scale, db = 1.0, None
                g = op.group
                if g is not None and op is not g.members[-1]:
                    pre = None                      # the group's ONE backward launch ran in front of its last member's data gradient
                else:
                    # (a group: the [M, K] tensors of all members, the gradient view addressed through the group's channel map)
                    lead = op if g is None else g.members[0]
                    Kb = K if g is None else g.K
                    y_ptr, dy_ptr = (op.y_view.ptr, op.dy_view.ptr) if g is None else (g.y.data_ptr(), g.dy.data_ptr())
                    fpool = getattr(op, "fused_pool", None)
                    if fpool is not None:
                        da = da_ptr = dmap = None
                    elif g is None:
                        da, da_ptr, dmap = self._gview(op.out), self._gview(op.out).ptr, None
                    else:
                        gviews = [self._gview(m.out) for m in g.members]
                        assert all(v.buf is gviews[0].buf and v.ld == gviews[0].ld for v in gviews)
                        base, cm = g.chan_map(gviews)
                        da, da_ptr, dmap = gviews[0], gviews[0].buf.data_ptr() + 2 * base, C.byref(cm)
                        self._keep = getattr(self, "_keep", []) + [cm]
                    mean, rstd = self._sl(self.bn_mean, lead.beta_off, Kb), self._sl(self.bn_rstd, lead.beta_off, Kb)
                    dbeta = self._sl(self.Btg, lead.beta_off, Kb)
                    rows = l.mbx_bn_bwd_rows(M, Kb)
                    beta = self._sl(self.Bt, lead.beta_off, Kb)
                    ws_off = lead.bn_ws_off

                    def pre_onepass(s, op=op, da=da, da_ptr=da_ptr, dmap=dmap, mean=mean, rstd=rstd, beta=beta, dbeta=dbeta, Kb=Kb, M=M,
                                    y_ptr=y_ptr, dy_ptr=dy_ptr, ws_off=ws_off):
                        # da and y are read once: the slice stays in registers across a grid barrier
                        _lib.check(l.mbx_bn_bwd_onepass_mapped(da_ptr, da.ld, int(op.relu), y_ptr, M, Kb, mean.data_ptr(),
                                                               rstd.data_ptr(), beta.data_ptr(), dbeta.data_ptr(),
                                                               dy_ptr, self.bn_ws.data_ptr() + 4 * ws_off,
                                                               self.cu_cap or self.bn_max_wg, self.step_ctl.data_ptr(), dmap, s),
                                   "bn_bwd_onepass")

                    def pre(s, op=op, da=da, da_ptr=da_ptr, dmap=dmap, mean=mean, rstd=rstd, beta=beta, dbeta=dbeta, rows=rows, Kb=Kb,
                            M=M, y_ptr=y_ptr, dy_ptr=dy_ptr):
                        # relu mask recomputed from y (a = NULL): the activation is not re-read in the backward pass
                        _lib.check(l.mbx_bn_bwd_reduce_mapped(da_ptr, da.ld, None, 0, int(op.relu), y_ptr, M, Kb,
                                                              mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(),
                                                              self.bwd_scratch.data_ptr(), dmap, s), "bn_bwd_reduce")
                        _lib.check(l.mbx_bn_bwd_finalize(self.bwd_scratch.data_ptr(), rows, Kb, M, dbeta.data_ptr(),
                                                         self.m12.data_ptr(), s), "bn_bwd_finalize")
                        _lib.check(l.mbx_bn_bwd_apply_mapped(da_ptr, da.ld, None, 0, int(op.relu), y_ptr, M, Kb,
                                                             mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), self.m12.data_ptr(),
                                                             dy_ptr, dmap, s), "bn_bwd_apply")
                    if op.bw_unit is not None:
                        # the sums were ADDED by the data gradient(s) that wrote da (their launches ran earlier in this pass)
                        thr = self._sl(self.bn_thr, lead.beta_off, Kb)
                        rows_off = lead.bw_rows_off

                        def pre(s, da=da, da_ptr=da_ptr, dmap=dmap, mean=mean, rstd=rstd, thr=thr, dbeta=dbeta, Kb=Kb, M=M,
                                y_ptr=y_ptr, dy_ptr=dy_ptr, rows_off=rows_off):
                            _lib.check(l.mbx_bn_bwd_apply_rows(self.bn_ws.data_ptr() + 4 * rows_off, STATS_ROWS, da_ptr, da.ld, y_ptr, M, Kb,
                                                               mean.data_ptr(), rstd.data_ptr(), thr.data_ptr(), dbeta.data_ptr(),
                                                               dy_ptr, dmap, s), "bn_bwd_apply_rows")
                    if fpool is not None:
                        gy = self._gview(fpool.out)
                        rows = l.mbx_bn_bwd_rows_pooled(fpool.x.N, fpool.x.H, fpool.x.W, Kb)     # (<= the plain form's: bwd_scratch fits)
                        assert 0 < rows * Kb * 2 <= self.bwd_scratch.numel()

                        def pre(s, op=op, gy=gy, fpool=fpool, mean=mean, rstd=rstd, beta=beta, dbeta=dbeta, rows=rows, Kb=Kb, M=M,
                                y_ptr=y_ptr, dy_ptr=dy_ptr):
                            geo = (gy.ptr, gy.img_stride, gy.ld, fpool.argmax.data_ptr(), fpool.x.N, fpool.x.H, fpool.x.W, fpool.out.H,
                                   fpool.out.W, int(op.relu), y_ptr, Kb, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr())
                            _lib.check(l.mbx_bn_bwd_reduce_pooled(*geo, self.bwd_scratch.data_ptr(), s), "bn_bwd_reduce_pooled")
                            _lib.check(l.mbx_bn_bwd_finalize(self.bwd_scratch.data_ptr(), rows, Kb, M, dbeta.data_ptr(),
                                                             self.m12.data_ptr(), s), "bn_bwd_finalize")
                            _lib.check(l.mbx_bn_bwd_apply_pooled(*geo, self.m12.data_ptr(), dy_ptr, s), "bn_bwd_apply_pooled")
                if pre is not None and op.bw_unit is None and (op.bn_ws_off if g is None else g.members[0].bn_ws_off) >= 0:
                    # chosen at CALL time: Trainer.check_health() falls back to the three launches (and re-captures its
                    # graphs) when a grid barrier has timed out -- e.g. RCCL kernels holding more CUs than bn_max_wg allows for
                    def pre(s, one=pre_onepass, three=pre):
                        (three if self.no_onepass else one)(s)
                    if os.environ.get("MBX_PROBE_BN_APPLY_ONLY") == "1":
                        # timing probe (WRONG results): what a batch-norm backward that is ONE streaming launch would cost --
                        # the apply launch of the three-launch form on stale totals
                        def pre(s, op=op, da=da, da_ptr=da_ptr, dmap=dmap, mean=mean, rstd=rstd, beta=beta, Kb=Kb, M=M, y_ptr=y_ptr,
                                dy_ptr=dy_ptr):
                            _lib.check(l.mbx_bn_bwd_apply_mapped(da_ptr, da.ld, None, 0, int(op.relu), y_ptr, M, Kb,
                                                                 mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), self.m12.data_ptr(),
                                                                 dy_ptr, dmap, s), "bn_bwd_apply")
            if op.kind == "bn" and op.fb_unit is not None and pre is not None:
                # this layer's backward runs as the tail of the data gradient(s) that wrote its activation gradient (their
                # launches came earlier in this pass): nothing to launch here -- unless the trainer fell back (call time)
                def pre(s, split=pre):
                    if self.no_onepass:
                        split(s)
                self.fused_bwd_layers += 1
            ddesc = fdesc = None
            if op.need_dx:
                gx = self._gview(op.x)
                acc, acc_src = self._claim(gx)
                dyin = View(dyv.buf, dyv.N, dyv.H, dyv.W, op.kpad, dyv.ld, dyv.ch_off, 2)
                mr = mask_of.get(id(op))
                ddesc = ops.make_desc(dyin, self.Wd[op.dgrad_off:], op.Cin, op.R, op.S, op.stride,
                                      op.R - 1 - op.pad_t, op.S - 1 - op.pad_l, gx, transposed=1, accumulate=acc,
                                      rscale=(scale if scale != 1.0 else 0.0), skip=(mr.out if mr is not None else None),
                                      acc_src=acc_src)
                if mr is not None and self.relu_bits and op.Cin % 8 == 0:
                    # the relu-backward mask from the SIGN BITS the residual launch writes beside its output (1/16 of the
                    # bytes of the bf16 tensor, on a launch that is bound by the tensors its epilogue streams); where the
                    # library has no such launch (the stride-2 data gradients of Mixed_6a / 7a) the bf16 form stays
                    bits = torch.zeros((op.x.N * op.x.H * op.x.W, (op.Cin + 31) // 32 * 4), dtype=torch.uint8, device=self.dev)
                    alt = ops.make_desc(dyin, self.Wd[op.dgrad_off:], op.Cin, op.R, op.S, op.stride,
                                        op.R - 1 - op.pad_t, op.S - 1 - op.pad_l, gx, transposed=1, accumulate=acc,
                                        rscale=(scale if scale != 1.0 else 0.0), acc_src=acc_src, relu_bits=bits)
                    if torch.device(self.dev).type != "cuda" or l.mbx_conv_supported(C.byref(alt)) == 0:
                        ddesc, mr.relu_bits = alt, bits
                self._tune(op, ddesc, "dgrad")
                if op.fb_segments is not None:
                    # this data gradient writes the activation gradient of batch-norm layers whose backward it runs as its tail
                    assert acc == 0 and acc_src is None and mr is None, op.name
                    fdesc = ops.ConvDesc.from_buffer_copy(ddesc)
                    fdesc.bn_bwd = C.addressof(self._fb_table(op))
                    fdesc.work_counter = None
                    if not (ops.I5_FLAG < fdesc.tile_config <= ops.I5_FLAG + 7 or fdesc.tile_config in (ops.DIRECTW_TILE_CONFIG, ops.RESIDENT_TILE_CONFIG)):
                        # the measured table chose a non-persistent tile: a grid-barrier launch needs every workgroup resident
                        fdesc.tile_config = ops.i5_tile_for(op.x.M, op.Cin, self.n_cus)
                    if l.mbx_conv_supported(C.byref(fdesc)) != 0:                      # (e.g. the 128 x 192 tile: no tail)
                        fdesc.tile_config = ops.i5_tile_for(op.x.M, op.Cin, self.n_cus)
                    _lib.check(l.mbx_conv_supported(C.byref(fdesc)), "data gradient + BN backward " + op.name)
                    self.fused_bwd_launches += 1
                if op.bw_segments is not None:
                    # this data gradient writes the activation gradient of batch-norm layers on the streaming backward: its
                    # epilogue adds their sums (a plain store by construction: _plan_bw_stats)
                    assert acc == 0 and acc_src is None and mr is None, op.name
                    ddesc.bn_bwd_stats = C.addressof(self._bw_table(op))
                    if l.mbx_conv_supported(C.byref(ddesc)) != 0:
                        # (the 256 x 128 persistent tile and the panel-resident launch have no statistics epilogue)
                        for alt in (ops.I5_TILE_CONFIGS[2], ops.I5_TILE_CONFIGS[1], 0):
                            ddesc.tile_config = alt
                            if alt == 0:
                                ddesc.work_counter = None
                            if l.mbx_conv_supported(C.byref(ddesc)) == 0:
                                break
                        _lib.check(l.mbx_conv_supported(C.byref(ddesc)), "statistics epilogue " + op.name)

            # the weight gradient is DEFERRED: one grouped launch per backward segment (make_wgrad_groups)
            job = ops.WgradJob()
            job.desc = ops.make_desc(op.x, None, K, op.R, op.S, op.stride, op.pad_t, op.pad_l,
                                     View(dyv.buf, op.out.N, op.out.H, op.out.W, 8, dyv.ld, dyv.ch_off, 2))
            job.desc.C_out = K
            job.dy, job.dy_img_stride, job.ld_dy = dyv.ptr, dyv.img_stride, dyv.ld
            job.scale = float(scale)
            job.dw, job.db = dw.data_ptr(), (None if db is None else db.data_ptr())

            grp = getattr(op, "group", None) if op.kind == "bn" else None
            if grp is not None and ddesc is not None:
                grp.dgrad_desc[grp.members.index(op)] = ddesc

            op.fdesc = fdesc

            def run(op=op, pre=pre, ddesc=ddesc, grp=grp, fdesc=fdesc):
                s = st()
                if pre is not None and not (op.kind == "bn" and self._probe_skip_bn_bwd):
                    pre(s)
                if fdesc is not None and not self.no_onepass:
                    fdesc.max_workgroups = self.cu_cap or self.bn_max_wg
                    _lib.check(l.mbx_conv(C.byref(fdesc), s), "dgrad + bn backward " + op.name)
                    return
                if ddesc is not None:
                    if grp is not None and grp.pair_bwd:
                        # both members' data gradients in ONE launch, issued with the last member (first in backward order)
                        if op is grp.members[-1]:
                            _lib.check(l.mbx_conv_pair(C.byref(ddesc), C.byref(grp.dgrad_desc[0]), s), "dgrad pair " + op.name)
                        return
                    ddesc.max_workgroups = self.cu_cap          # (persistent launches; baked into a captured graph)
                    _lib.check(l.mbx_conv(C.byref(ddesc), s), "dgrad " + op.name)
            L.append(run)
            self.bwd_ops.append(op)
            self.bwd_jobs.append(job)
        if torch.device(self.dev).type == "cuda" and os.environ.get("MBX_CONV_PAIR", "1") != "0":
            for grp in self.bn_groups:
                if len(grp.members) == 2 and len(grp.dgrad_desc) == 2 and not any(getattr(m, "fdesc", None) is not None for m in grp.members):
                    grp.pair_bwd = l.mbx_conv_pair(C.byref(grp.dgrad_desc[1]), C.byref(grp.dgrad_desc[0]), st()) == 0
        assert not self.pending_acc, "a residual block's input gradient was never written"
        self.bwd_launches = L
        self.fwd_launches = self._build_forward_launches()
        self._default_groups = None

    def _pool_producer(self, pool):
        """The batch-norm layer whose activation is `pool`'s input if the pool's backward can ride in that layer's three-launch
        BN backward (3x3 / 2 VALID max-pool; the activation has no other reader; the layer is not in a group and too large
        for the one-launch form: the two stem pools at BATCH_SIZE 64), else None.  MBX_POOL_FUSE=0 turns it off (A/B)."""
        if os.environ.get("MBX_POOL_FUSE", "1") == "0" or pool.kind != "max" or pool.k != 3 or pool.stride != 2 or pool.pad != 0 \
                or pool.argmax is None:
            return None
        readers = [o for o in self.fwd if o is not pool and o.x.buf is pool.x.buf]
        if readers:
            return None
        for c in self.convs:
            if c.out.buf is pool.x.buf and c.out.ch_off == pool.x.ch_off and c.out.C == pool.x.C and c.out.ld == pool.x.ld == pool.x.C:
                # (never on the one-launch backward: too large for it -- or the deterministic mode, which uses the three launches
                # everywhere and for good, so that the choice does not depend on a grid cap)
                ok = c.kind == "bn" and c.trainable and c.group is None and (c.bn_ws_off < 0 or self.deterministic)
                return c if ok else None
        return None

    def make_wgrad_groups(self, job_lists):
        """One grouped weight-gradient launch (ops.WgradGroup) per list of jobs; the caller runs each after the
        backward launches its jobs belong to (Trainer: at the end of every backward segment)."""
        return [ops.WgradGroup(jobs, deterministic=self.deterministic, device=self.dev) for jobs in job_lists if jobs]

    # --------------------------------------------------------------------- running
    def prepare_filters(self):
        """Refresh the dgrad filter copies from the bf16 filters (once per step, after the update)."""
        if self.mode == "train" and self.filter_entries:
            _lib.check(_lib.lib().mbx_filter_prepare(self.Wb.data_ptr(), self.Wd.data_ptr(), self.filter_table.data_ptr(),
                                                     self.filter_entries, self.filter_blocks,
                                                     torch.cuda.current_stream().cuda_stream), "filter_prepare")

    def apply_moving_update(self, skip_ctl=None, skipped_steps=None, ema_mean=None, ema_var=None, ema_decay=0.0):
        """The deferred moving-average update of every training-mode batch-norm layer (defer_moving), one launch.
        skip_ctl / skipped_steps: device tensors (step control block; int64 counter of skipped steps) or None.
        ema_mean / ema_var: the EMA shadows of the moving statistics ([nBt] each), updated from the new values in the same launch."""
        lo = self.head_bt_start if self.fine_tune else 0          # --fine_tune: the backbone's batch norm is frozen (train.py:124-131)
        n = self.nBt - lo
        if n > 0 and self.mode == "train":
            _lib.check(_lib.lib().mbx_bn_moving_update(self.MM.data_ptr() + 4 * lo, self.MV.data_ptr() + 4 * lo,
                                                       self.bn_mean.data_ptr() + 4 * lo, self.bn_var.data_ptr() + 4 * lo, n,
                                                       self.bn_decay, ops._p(skip_ctl), ops._p(skipped_steps),
                                                       None if ema_mean is None else ema_mean.data_ptr() + 4 * lo,
                                                       None if ema_var is None else ema_var.data_ptr() + 4 * lo, float(ema_decay),
                                                       torch.cuda.current_stream().cuda_stream), "bn_moving_update")

    def fold_bn(self):
        """Frozen BN -> per-channel scale/shift for the conv epilogue (detect.py:313-326)."""
        _lib.check(_lib.lib().mbx_bn_fold(self.MM.data_ptr(), self.MV.data_ptr(), self.Bt.data_ptr(), BN_EPS, self.nBt,
                                          self.bn_scale.data_ptr(), self.bn_shift.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "bn_fold")

    def set_input(self, images_f32):
        """images [B,S,S,3] float32 in [-1,1] (inputs.py:350-351) -> packed bf16 NHWC8."""
        assert images_f32.shape == (self.B, self.S, self.S, 3) and images_f32.dtype == torch.float32
        _lib.check(_lib.lib().mbx_pack_input(images_f32.contiguous().data_ptr(), self.B * self.S * self.S,
                                             self.images.buf.data_ptr(), torch.cuda.current_stream().cuda_stream), "pack_input")

    def forward(self):
        if self._i5_used or self.atomic_stats:
            self._fwd_clear.zero_()                # work counters (forward AND backward launches of this pass) + statistics rows
        for f in self.fwd_launches:
            f()
        return self.locs, self.logits

    def layer_table(self):
        """The network as the reference states it (model.py:6-337): one entry per slim.conv2d / pool in forward
        order, fused sibling launches expanded into their members, inputs resolved to the scopes that produced the
        channels read.  tests/test_model_graph.py compares it with the table generated from the reference's source."""
        writers = {}                       # id(buffer) -> [(ch_lo, ch_hi, scope)] latest writer per channel range

        def record(v: View, scope):
            lst = [w for w in writers.get(id(v.buf), []) if w[1] <= v.ch_off or w[0] >= v.ch_off + v.C]
            lst.append((v.ch_off, v.ch_off + v.C, scope))
            writers[id(v.buf)] = lst

        def producers(v: View):
            if v.buf is self.images.buf:
                return ["inputs"]
            inside = sorted(w for w in writers.get(id(v.buf), []) if w[0] >= v.ch_off and w[1] <= v.ch_off + v.C)
            assert inside and inside[0][0] == v.ch_off and inside[-1][1] == v.ch_off + v.C and \
                all(a[1] == b[0] for a, b in zip(inside, inside[1:])), "input view is not tiled by earlier outputs"
            return [w[2] for w in inside]

        table = []
        for op in self.fwd:
            if isinstance(op, PoolOp):
                table.append({"scope": op.scope, "op": op.kind + "_pool2d", "in_channels": op.x.C, "out_channels": op.out.C,
                              "kernel": [op.k, op.k], "stride": op.stride, "pad": [op.pad, op.pad],
                              "out_hw": [op.out.H, op.out.W], "inputs": producers(op.x)})
                record(op.out, op.scope)
                continue
            ins = producers(op.x)
            off = 0
            for m in op.members:
                e = {"scope": m.scope, "op": "conv2d", "in_channels": 3 if op.x.buf is self.images.buf else op.Cin,
                     "out_channels": m.K, "kernel": [op.R, op.S], "stride": op.stride, "pad": [op.pad_t, op.pad_l],
                     "out_hw": [op.out.H, op.out.W], "inputs": ins, "bn": op.kind in ("bn", "frozen"),
                     "bias": op.kind == "residual", "activation": "relu" if (op.relu and op.kind != "residual") else None}
                if op.kind == "residual":
                    e["residual"] = {"scale": op.rscale, "skip": producers(op.skip), "activation": "relu" if op.relu else None}
                table.append(e)
                if op.kind != "head":
                    record(op.out.slice(off, m.K), m.scope)
                off += m.K
        return table

    def flops_per_image(self, train=True):
        """Algorithmic FLOPs per image (2 per multiply-add) of the convolutions (SURVEY 8(d) convention): forward
        only, or a training step counted as 3x forward for trainable layers (dgrad + wgrad) and 1x for frozen ones."""
        total = 0.0
        for op in self.convs:
            f = 2.0 * (op.M // self.B) * op.K * op.Cin * op.R * op.S
            total += f * (3.0 if (train and op.trainable) else 1.0)
        return total

    def barrier_timeouts(self):
        """Workgroups of one-launch BN-backward launches that gave up on their grid barrier (the grid was not resident)
        SINCE THE NET WAS BUILT: word [0] of the step control block (every such workgroup adds 1; data-parallel runs sum it
        over ranks with the gradients) is folded into a running total before it is cleared at the start of the next
        backward pass, so a timeout between two health checks is not lost.  0 in a healthy run.  Host sync."""
        if self.mode != "train":
            return 0
        return int(self.bn_timeouts_total) + int(float(self.step_ctl[0]))

    def zero_grads(self):
        """ONE launch (mbx_step_begin): Wg, Btg, the step control block, the accumulators / arrival counters of the
        one-launch BN backward and the optimiser's regularisation-loss accumulator cleared; the control word's time-out
        count kept in bn_timeouts_total first."""
        # --fine_tune: only the heads' filter gradients are ever written (and all-reduced: Trainer.w_lo): the frozen backbone's
        # 218 MB of the buffer are not cleared every step (39 -> ~10 us of a 4.4 ms step); the launch clears the beta gradients
        # and the control block in front of them, a fill the heads' range
        n_clear = self.G_off if self.fine_tune else self.G.numel()
        _lib.check(_lib.lib().mbx_step_begin(self.G.data_ptr(), n_clear, self.bn_ws.data_ptr(), self.bn_ws.numel(),
                                             self.nBt, self.bn_timeouts_total.data_ptr(), self.reg_loss.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream), "step_begin")
        if self.fine_tune:
            self.Wg[self.head_w_start:].zero_()

    def backward(self):
        """d_locs / d_logits must hold the loss gradients; fills Wg / Btg.  (Eager form: all data-gradient launches,
        then the deferred weight gradients in four grouped launches; the Trainer interleaves them per segment.)"""
        if self._i5_used:
            self.i5_counters.zero_()
        for f in self.bwd_launches:
            f()
        self.run_deferred_wgrad()

    def run_deferred_wgrad(self):
        """The weight gradients of every layer, after ALL of bwd_launches have run (their dy / x are kept alive)."""
        if self._default_groups is None:
            jobs = [j for j in self.bwd_jobs if j is not None]
            n = max(1, (len(jobs) + 3) // 4)
            self._default_groups = self.make_wgrad_groups([jobs[i:i + n] for i in range(0, len(jobs), n)])
        for g in self._default_groups:
            g.launch()


class PoolOp:
    def __init__(self, net, kind, x: View, out: View, k, stride, pad):
        self.net, self.kind, self.x, self.out, self.k, self.stride, self.pad = net, kind, x, out, k, stride, pad
        self.argmax = None
        if kind == "max" and net.mode == "train" and not net.fine_tune:
            self.argmax = torch.zeros((out.N, out.H, out.W, out.C), dtype=torch.uint8, device=net.dev)

    def needs_backward(self):
        return self.net.mode == "train" and not self.net.fine_tune

    def forward(self):
        l, x, y = _lib.lib(), self.x, self.out
        s = torch.cuda.current_stream().cuda_stream
        if self.kind == "max":
            _lib.check(l.mbx_maxpool_fwd(x.ptr, x.img_stride, x.ld, x.N, x.H, x.W, x.C, self.k, self.stride, y.ptr,
                                         y.img_stride, y.ld, y.H, y.W, None if self.argmax is None else self.argmax.data_ptr(), s),
                       "maxpool_fwd")
        else:
            _lib.check(l.mbx_avgpool_fwd(x.ptr, x.img_stride, x.ld, x.N, x.H, x.W, x.C, self.k, self.pad, y.ptr,
                                         y.img_stride, y.ld, y.H, y.W, s), "avgpool_fwd")

    def make_backward(self, gy: View, gx: View, acc):
        l, x, y = _lib.lib(), self.x, self.out

        def run():
            s = torch.cuda.current_stream().cuda_stream
            if self.kind == "max":
                _lib.check(l.mbx_maxpool_bwd(gy.ptr, gy.img_stride, gy.ld, self.argmax.data_ptr(), x.N, x.H, x.W, x.C, self.k,
                                             self.stride, y.H, y.W, gx.ptr, gx.img_stride, gx.ld, acc, s), "maxpool_bwd")
            else:
                _lib.check(l.mbx_avgpool_bwd(gy.ptr, gy.img_stride, gy.ld, x.N, x.H, x.W, x.C, self.k, self.pad, y.H, y.W,
                                             gx.ptr, gx.img_stride, gx.ld, acc, s), "avgpool_bwd")
        return run
