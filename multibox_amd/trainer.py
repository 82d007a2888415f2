"""One training step of the reference's train.py:173-298 on the libmbx engine.

step = forward (model.build) -> decode+match+loss (loss.add_loss) -> backward -> [data-parallel
gradient all-reduce, SUM: the reference loss is a batch sum, loss.py:100-101] -> RMSProp + L2 +
EMA (train.py:190-263) -> refresh of the bf16 / dgrad filter copies.

The forward+loss+backward launch sequence is captured into hipGraphs (one per backward
segment so that each segment's gradient bucket can be all-reduced over RCCL/xGMI while the next
segment computes); the optimizer runs eagerly because its learning rate / EMA decay change
every step.
"""
from __future__ import annotations

import math

import torch

from . import _lib
from .engine import Net, WEIGHT_DECAY
from .loss import MultiboxLoss
from .dist import BucketReducer


def decay_steps(num_train_examples, batch_size, num_epochs_per_decay):
    """train.py:193-195 (Python-2 integer division)."""
    return int((int(num_train_examples) // int(batch_size)) * num_epochs_per_decay)


def learning_rate(step, lr0, dsteps, factor, staircase=True):
    """tf.train.exponential_decay as called at train.py:198-204."""
    p = step / float(dsteps)
    if staircase:
        p = math.floor(p)
    return float(lr0) * float(factor) ** p


class Trainer:
    def __init__(self, net: Net, bbox_priors, max_num_bboxes=13, location_loss_alpha=1000.0, initial_learning_rate=0.01,
                 decay_steps_=7116, learning_rate_decay_factor=0.94, staircase=True, rmsprop_decay=0.9,
                 rmsprop_momentum=0.0, rmsprop_epsilon=1.0, moving_average_decay=0.9999, use_graph=True,
                 n_segments=None, process_group=None):
        assert net.mode == "train"
        self.net = net
        self.loss = MultiboxLoss(bbox_priors, net.B, max_num_bboxes, location_loss_alpha, device=net.dev)
        assert self.loss.P == net.P, "priors (%d) do not match the network's predictions (%d)" % (self.loss.P, net.P)
        self.loss.d_locs, self.loss.d_logits = net.d_locs, net.d_logits
        self.lr0, self.dsteps, self.lr_factor, self.staircase = initial_learning_rate, decay_steps_, learning_rate_decay_factor, staircase
        self.rms_decay, self.momentum, self.eps, self.ema_decay = rmsprop_decay, rmsprop_momentum, rmsprop_epsilon, moving_average_decay
        self.global_step = 0
        self.reducer = BucketReducer(process_group)
        f32 = dict(dtype=torch.float32, device=net.dev)
        # trainable ranges of the flat buffers (heads are last in forward order)
        self.w_lo = net.head_w_start if net.fine_tune else 0
        self.bt_lo = net.head_bt_start if net.fine_tune else 0
        self.Wms = torch.ones(net.nW, **f32)            # TF initialises the rms slot to ones
        self.Btms = torch.ones(net.nBt, **f32)
        self.Wmom = torch.zeros(net.nW, **f32) if self.momentum != 0.0 else None
        self.Btmom = torch.zeros(net.nBt, **f32) if self.momentum != 0.0 else None
        # EMA shadows of every model variable, incl. BN moving statistics (train.py:257)
        self.Wema, self.Btema = net.W.clone(), net.Bt.clone()
        self.MMema, self.MVema = net.MM.clone(), net.MV.clone()
        self.frozen_reg = 0.0
        if self.w_lo > 0:       # regulariser of frozen variables is a constant (train.py:246 still adds it)
            self.frozen_reg = float(0.5 * WEIGHT_DECAY * (net.W[:self.w_lo].double() ** 2).sum())
        self.gt = torch.zeros((net.B, max_num_bboxes, 4), **f32)
        self.n_gt = torch.zeros((net.B,), dtype=torch.int32, device=net.dev)
        self.images = torch.zeros((net.B, net.S, net.S, 3), **f32)
        net.prepare_filters()
        if net.fine_tune:
            net.fold_bn()
        self.use_graph = use_graph
        self.graphs = None
        # backward segments = gradient buckets: data-parallel runs use more of them, so that the bucket that can only
        # start after the LAST backward launch (nothing left to overlap it with) is small
        if n_segments is None:
            n_segments = 6 if self.reducer.enabled else 4
        self._segments = self._make_segments(n_segments)

    # ------------------------------------------------------------------ segments / buckets
    def _make_segments(self, n):
        """Split the backward launch list into n runs of roughly equal parameter count; each run's
        gradients are one contiguous bucket of Wg (backward order = reverse parameter order)."""
        net = self.net
        ops_rev = [op for op in reversed(net.fwd)]
        launches = net.bwd_launches
        # map launches to ops: bwd_launches was built over reversed(net.fwd) skipping some ops
        tagged = []
        it = iter(launches)
        from .engine import PoolOp
        for op in ops_rev:
            if isinstance(op, PoolOp):
                if op.needs_backward():
                    tagged.append((next(it), None))
            elif op.trainable:
                tagged.append((next(it), op))
        total = net.nW - self.w_lo
        target = max(total // max(n, 1), 1)
        segs, cur, hi = [], [], net.nW
        lo = hi
        for fn, op in tagged:
            cur.append(fn)
            if op is not None:
                lo = min(lo, op.w_off)
                if hi - lo >= target and len(segs) < n - 1:
                    segs.append((cur, lo, hi))
                    cur, hi = [], lo
        if cur:
            segs.append((cur, self.w_lo, hi))
        return segs

    # ---------------------------------------------------------------------------- step
    def _front(self):
        net = self.net
        net.set_input(self.images)
        net.forward()
        self.loss.forward_backward(net.locs, net.logits, self.gt, self.n_gt)
        net.zero_grads()

    def _capture(self):
        """Warm up eagerly once (lazy module loads, hipFuncSetAttribute), then capture."""
        self._front()
        for fns, _, _ in self._segments:
            for f in fns:
                f()
        torch.cuda.synchronize()
        graphs = []
        # thread_local: RCCL's watchdog thread may query events while we capture (data-parallel runs)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._front()
        graphs.append(g)
        for fns, _, _ in self._segments:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for f in fns:
                    f()
            graphs.append(g)
        self.graphs = graphs

    def set_batch(self, images, gt_bboxes, num_gt_bboxes):
        self.images.copy_(images)
        self.gt.copy_(gt_bboxes)
        self.n_gt.copy_(num_gt_bboxes)

    def step(self):
        """One optimisation step on the batch set by set_batch().  Returns nothing; read
        self.loss.loss2 / self.total_loss() when needed (no host sync here)."""
        net, red = self.net, self.reducer
        if self.use_graph:
            if self.graphs is None:
                self._capture()
            self.graphs[0].replay()
            for g, (_, lo, hi) in zip(self.graphs[1:], self._segments):
                g.replay()
                red.reduce_async(net.Wg, lo, hi)
        else:
            self._front()
            for fns, lo, hi in self._segments:
                for f in fns:
                    f()
                red.reduce_async(net.Wg, lo, hi)
        red.reduce_async(net.Btg, self.bt_lo, net.nBt)
        red.wait()
        self._optimizer()
        self.global_step += 1

    def run_eager_once(self):
        """forward + loss + backward launched eagerly (profiling aid; no optimizer, no all-reduce)."""
        self._front()
        for fns, _, _ in self._segments:
            for f in fns:
                f()

    def _optimizer(self):
        net, l = self.net, _lib.lib()
        s = torch.cuda.current_stream().cuda_stream
        t = self.global_step
        lr = learning_rate(t, self.lr0, self.dsteps, self.lr_factor, self.staircase)
        d = min(self.ema_decay, (1.0 + t) / (10.0 + t))        # ExponentialMovingAverage(num_updates=global_step)
        self.lr = lr
        net.reg_loss.zero_()
        P = lambda t_, off=0: None if t_ is None else t_.data_ptr() + 4 * off
        lo, n = self.w_lo, net.nW - self.w_lo
        _lib.check(l.mbx_rmsprop_ema_step(P(net.W, lo), P(net.Wg, lo), P(self.Wms, lo), P(self.Wmom, lo), P(self.Wema, lo),
                                          net.Wb.data_ptr() + 2 * lo, n, lr, self.rms_decay, self.momentum, self.eps,
                                          WEIGHT_DECAY, d, 1, net.reg_loss.data_ptr(), s), "rmsprop W")
        lo, n = self.bt_lo, net.nBt - self.bt_lo
        _lib.check(l.mbx_rmsprop_ema_step(P(net.Bt, lo), P(net.Btg, lo), P(self.Btms, lo), P(self.Btmom, lo), P(self.Btema, lo),
                                          None, n, lr, self.rms_decay, self.momentum, self.eps, 0.0, d, 1, None, s), "rmsprop beta")
        _lib.check(l.mbx_ema_update(P(self.MMema, lo), P(net.MM, lo), n, d, s), "ema moving_mean")
        _lib.check(l.mbx_ema_update(P(self.MVema, lo), P(net.MV, lo), n, d, s), "ema moving_var")
        net.prepare_filters()

    def losses(self):
        """(location_loss, confidence_loss, regularization_loss, total_loss) -- host sync."""
        l2 = self.loss.loss2.tolist()
        reg = float(self.net.reg_loss) + self.frozen_reg
        return l2[0], l2[1], reg, l2[0] + l2[1] + reg

    def match_status(self):
        return self.loss.status
