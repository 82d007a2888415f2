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
import os
import re

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
                 n_segments=None, process_group=None, trainable_scopes=None):
        assert net.mode == "train"
        self.net = net
        self.pg = process_group
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
        self.refresh_frozen_reg()
        # --trainable_scopes (train.py:152-171): only variables under the given scopes are updated; everything else in
        # the trainable range still gets its EMA / regulariser / bf16 refresh (it stays a model variable)
        self.opt_ranges = self._optimizer_ranges(trainable_scopes)
        self.gt = torch.zeros((net.B, max_num_bboxes, 4), **f32)
        self.n_gt = torch.zeros((net.B,), dtype=torch.int32, device=net.dev)
        self.images = torch.zeros((net.B, net.S, net.S, 3), **f32)
        net.prepare_filters()
        if net.fine_tune:
            net.fold_bn()
        self.use_graph = use_graph
        self.graphs = None
        self.events = []                               # things a log should show (BN-backward fall-back, ...)
        self._timeouts_seen = 0
        self._stop_flag = torch.zeros((), **f32)       # request_stop(): copied into the step control block every step
        # A skipped step (control word non-zero) is skipped EVERYWHERE: the forward pass only records the batch statistics
        # and the moving averages are updated in the optimiser phase behind the same gate (Net.apply_moving_update), which
        # also counts the skipped steps; check_health() takes them off global_step again (the learning-rate schedule,
        # NUM_TRAIN_ITERATIONS and checkpoint names count APPLIED steps).
        net.defer_moving = True
        self.skipped_steps = torch.zeros((), dtype=torch.int64, device=net.dev)
        self._skipped_seen = 0
        # backward segments = gradient buckets (each also ends in one grouped weight-gradient launch and is one hipGraph).
        # Four of 60 MB; data-parallel runs cut the LAST one once more: the 2 M parameters at the bottom of the network
        # (stem, Mixed_5b, block35: 6.7 MB of the 240) get a bucket of their own, so that the all-reduce nothing overlaps
        # moves 7 MB instead of 59 (the bucket before it -- the lower half of block17 and Mixed_6a -- is handed over with a
        # quarter of the backward pass still to run).  More, smaller buckets measured slower (each segment costs ~0.04 ms:
        # 6 + 1 segments 17.85 ms, 4 + 1 17.65, one rank with MBX_FORCE_DIST=1) and hide nothing more: at 2 ranks on one
        # xGMI link a 60 MB bucket takes ~0.9 ms and the next one is ~2 ms of backward away.
        tail = 0
        # OVERLAPPED weight gradients (net.wgrad_overlap_cus > 0, captured steps): the grouped launch of segment i runs on a
        # second stream with a capped grid beside the chain of the segments behind it (_capture).  Without a process group
        # the segments are only weight-gradient groups: more of them, cut by weight-gradient work instead of parameters, so
        # that the first group starts early and the last one -- the only one nothing overlaps -- is small.
        self.overlap_cus = net.wgrad_overlap_cus if (use_graph and torch.device(net.dev).type == "cuda") else 0
        by_work = False
        if n_segments is None:
            # without a process group the segments are not buckets: ONE backward segment = one grouped weight-gradient launch
            # for the whole network (longest-first over every layer, one launch tail instead of four) and one captured graph
            # for the whole step -- 16.54 -> 16.29 ms against four segments, same box (8 segments: 16.78)
            # (round 5: TWO buckets of ~117 MB + the 6.7 MB tail instead of four of 60 MB.  On one rank, collectives switched off
            # (MBX_DP_NO_ALLREDUCE=1), the bucketed form of the step costs +0.03 ms with one segment, +0.15 with 2 + tail, +0.25 with
            # 4 + tail against the single-GPU step -- each segment is one more grouped weight-gradient launch and one more graph;
            # by the xGMI estimate of dist.py a 117 MB bucket takes ~0.8 ms at 8 ranks and the first is ready with half of the
            # backward pass still to run: tools/dp_ab.sh, profiles/r05_dp_form_ab.txt)
            n_segments = int(os.environ.get("MBX_DP_SEGMENTS", "2")) if self.reducer.enabled else int(os.environ.get("MBX_SEGMENTS", "1"))
            tail = int(os.environ.get("MBX_DP_TAIL_PARAMS", "2000000")) if self.reducer.enabled else 0      # (A/B knobs)
            if self.overlap_cus and not self.reducer.enabled:
                n_segments, by_work = int(os.environ.get("MBX_WG_GROUPS", "8")), True
        self._side = None
        self.exposed_events = None                     # a list: step() brackets reducer.wait() with an event pair (bench.py)
        self._segments = self._make_segments(n_segments, tail_params=tail, by_work=by_work)

    def refresh_frozen_reg(self):
        """Regulariser of the frozen backbone under --fine_tune: a constant that train.py:246 still adds to the total
        loss.  Recomputed after every restore (the weights it is taken from change there)."""
        self.frozen_reg = 0.0
        if self.w_lo > 0:
            self.frozen_reg = float(0.5 * WEIGHT_DECAY * (self.net.W[:self.w_lo].double() ** 2).sum())

    def _optimizer_ranges(self, scopes):
        """[(buffer 'W' | 'Bt', lo, hi, trainable)] covering the trainable part of both flat buffers, adjacent ranges of
        equal status merged.  scopes=None: everything trainable (train.py:160-161).  A scope selects the variables
        whose name it matches from the start, like tf.get_collection(TRAINABLE_VARIABLES, scope) (train.py:168)."""
        net = self.net
        if scopes is None:
            return [("W", self.w_lo, net.nW, 1), ("Bt", self.bt_lo, net.nBt, 1)]
        pats = [re.compile(s.strip()) for s in scopes]
        marks = {"W": [], "Bt": []}
        self.trainable_names = []
        for name, (buf, off, shape, cpad) in net.param_index.items():
            if buf not in marks:
                continue                                         # moving statistics are not trainable variables
            n = 1
            for d in (shape[:-1] + (cpad,) if cpad is not None else shape):
                n *= d
            lo_buf = self.w_lo if buf == "W" else self.bt_lo
            if off < lo_buf:
                continue                                         # frozen by --fine_tune
            on = int(any(p.match(name) for p in pats))
            if on:
                self.trainable_names.append(name)
            marks[buf].append((off, off + n, on))
        out = []
        for buf, lo_buf, hi_buf in (("W", self.w_lo, net.nW), ("Bt", self.bt_lo, net.nBt)):
            cur_lo, cur_on = lo_buf, None
            pos = lo_buf
            for lo, hi, on in sorted(marks[buf]):
                # alignment gaps between variables ride with the range before them (their gradient is zero)
                if cur_on is None:
                    cur_on = on
                elif on != cur_on:
                    out.append((buf, cur_lo, lo, cur_on))
                    cur_lo, cur_on = lo, on
                pos = hi
            if cur_on is not None:
                out.append((buf, cur_lo, hi_buf, cur_on))
        return out

    def broadcast_parameters(self, src=0):
        """Data-parallel start: every rank takes rank `src`'s variables, optimiser slots and EMA shadows (after a
        restore only rank 0's files need to exist; identical seeds are not relied upon)."""
        if self.pg is None:
            return
        import torch.distributed as dist
        net = self.net
        for t in (net.W, net.Bt, net.MM, net.MV, self.Wms, self.Btms, self.Wmom, self.Btmom, self.Wema, self.Btema,
                  self.MMema, self.MVema):
            if t is not None:
                dist.broadcast(t, src=src, group=self.pg)
        step = torch.tensor([self.global_step], dtype=torch.int64, device=net.W.device)
        dist.broadcast(step, src=src, group=self.pg)
        self.global_step = int(step)
        if net.W.is_cuda:
            net.refresh_bf16()
            if net.fine_tune:
                net.fold_bn()
        self.refresh_frozen_reg()

    def request_stop(self):
        """Data-parallel runs: this rank cannot go on (its input is exhausted).  Raises word [1] of the step control block;
        from the next step on the optimiser launches of EVERY rank skip the update (the word is summed over ranks with
        the beta gradients) and check_health() reports `stop` on every rank at the next logging interval -- no per-step
        host synchronisation (train.py used to all-reduce an `ok` flag and read it back before every step)."""
        self._stop_flag.fill_(1.0)

    def fold_skipped(self, **note):
        """Take the steps the optimiser did NOT apply (flagged step control word: counted on the device, the same number on
        every rank) off global_step -- one host read of the counter, no collective.  check_health() does it every
        LOG_EVERY_N_STEPS; train.py also calls it in front of every checkpoint save, so that a checkpoint never names or
        stores a step count that includes unapplied steps.  Returns the number folded."""
        sk = getattr(self, "skipped_steps", None)
        skipped = (int(sk) - self._skipped_seen) if sk is not None else 0
        if skipped:
            self._skipped_seen += skipped
            ev = {"event": "skipped_steps", "count": skipped, "global_step_before": self.global_step,
                  "global_step": self.global_step - skipped}
            ev.update(note)
            self.events.append(ev)
            self.global_step -= skipped                # only applied steps count
        return skipped

    def applied_global_step(self):
        """global_step without the steps the optimiser skipped since the last fold -- a host read of the device counter that
        changes NOTHING (rank 0's time-triggered checkpoint must not move its global_step ahead of the other ranks': the
        health check that folds the count is a collective every rank enters at the same global_step)."""
        sk = getattr(self, "skipped_steps", None)
        return self.global_step - ((int(sk) - self._skipped_seen) if sk is not None else 0)

    def check_health(self):
        """Host sync; call it at logging intervals.  Verdict over ALL ranks on the steps since the last call:
          * matching failed (non-finite predictions / more boxes than predictions; the reference's py_func raises there,
            loss.py:82) -> raises;
          * a grouped weight-gradient launch did not process every work item (stale queue heads after an aborted
            launch would make every later launch compute NOTHING, silently) -> re-uploads the plan image and raises;
          * a grid barrier of the one-launch batch-norm backward timed out (workgroups not co-resident: another
            stream's kernels held more CUs than bn_max_workgroups left).  Those steps poisoned their gradients AND
            raised word [0] of the step control block, so the optimiser did not apply them on any rank.  The trainer
            falls back to the three-launch BN backward (no grid barrier), drops its captured graphs (re-captured in
            this process by the next step()) and carries on; the event is recorded in self.events.  Raises only if
            timeouts are seen after the fallback.
        Steps the optimiser skipped since the last call (counted on the device by the gated moving-statistics launch; the
        same number on every rank, the control word being summed over ranks) are taken off global_step and recorded in
        self.events.  A stop request is evaluated FIRST: the steps behind it were not applied, so whatever the repeated
        (or never set) batch of the exhausted rank did to the matcher does not turn a clean end of input into an error.
        Returns {"stop": a rank called request_stop(), "fallback": this call switched the BN backward}."""
        net = self.net
        wg_bad = sum(0 if g.completed_ok() else 1 for g in getattr(self, "wgrad_groups", []))
        bad = torch.tensor([net.barrier_timeouts() - self._timeouts_seen, int(self.match_status().max() != 0), wg_bad,
                            int(float(net.step_ctl[1]) != 0.0 or float(self._stop_flag) != 0.0)], dtype=torch.int32, device=net.W.device)
        if self.pg is not None:
            import torch.distributed as dist
            if dist.get_world_size(self.pg) > 1:
                dist.all_reduce(bad, op=dist.ReduceOp.SUM, group=self.pg)
        t, m, w, stop = (int(v) for v in bad.tolist())
        self.fold_skipped(stop_requested=bool(stop), barrier_timeouts=t)
        if stop:
            return {"stop": True, "fallback": False}
        out = {"stop": bool(stop), "fallback": False}
        if t:
            # (FIRST: a timed-out barrier of a fused convolution + BN-apply launch poisons the ACTIVATIONS, so the same step's
            # matching saw non-finite predictions -- a consequence of the time-out, not an error of its own; the step was not
            # applied either way)
            self._timeouts_seen = net.barrier_timeouts()
            if net.no_onepass:
                raise RuntimeError("batch-norm backward: %d grid-barrier timeout(s) AFTER the fall-back to the three-launch form" % t)
            import sys
            net.no_onepass = True                      # backward launches pick their BN form at call time (engine.py)
            self.graphs = None                         # re-captured by the next step(), in this process
            self.events.append({"event": "bn_backward_fallback", "global_step": self.global_step, "barrier_timeouts": t})
            print("[multibox_amd] step %d: %d grid-barrier timeout(s) in the one-launch BN backward (workgroups not co-resident); "
                  "the affected steps were NOT applied; continuing with the three-launch BN backward" % (self.global_step, t),
                  file=sys.stderr)
            out["fallback"] = True
            return out
        if m:
            raise RuntimeError("bipartite matching failed on %d rank(s) (non-finite predictions or n_gt > P)" % m)
        if w:
            for g in self.wgrad_groups:
                g.reset()
            raise RuntimeError("grouped weight gradient: %d launch group(s) did not process all their work items "
                               "(queue heads re-uploaded)" % w)
        return out

    # ------------------------------------------------------------------ segments / buckets
    def _make_segments(self, n, tail_params=0, by_work=False):
        """Split the backward launch list into n runs of roughly equal parameter count; each run's
        gradients are one contiguous bucket of Wg (backward order = reverse parameter order).  Every run has
        ONE grouped weight-gradient launch for its layers (the weight gradients are deferred: ops.WgradGroup;
        self._seg_groups[i], None where a run has no trainable layer).
        tail_params > 0: the last run is split once more where at most that many parameters remain (n + 1 runs).
        by_work: cut by the layers' weight-gradient work (M x C_out x R S C_in) instead of by parameters.
        Returns [(chain launches, lo, hi)]."""
        net = self.net
        tagged = list(zip(net.bwd_launches, net.bwd_ops, net.bwd_jobs))
        from .engine import PoolOp
        total = net.nW - self.w_lo
        target = max(total // max(n, 1), 1)

        def work(job):
            d = job.desc
            return float(d.N) * d.H_out * d.W_out * d.C_out * d.R * d.S * d.C_in
        wtarget = max(sum(work(j) for j in net.bwd_jobs if j is not None) / max(n, 1), 1.0)
        wacc = 0.0
        segs, cur, jobs, hi = [], [], [], net.nW
        lo = hi
        for fn, op, job in tagged:
            cur.append(fn)
            if job is not None:
                jobs.append(job)
                wacc += work(job)
            if not isinstance(op, PoolOp):
                lo = min(lo, op.w_off)
                full = (wacc >= wtarget * (len(segs) + 1)) if by_work else (hi - lo >= target)
                if full and hi > lo and len(segs) < n - 1:
                    segs.append((cur, lo, hi, jobs))
                    cur, jobs, hi = [], [], lo
                elif tail_params > 0 and len(segs) == n - 1 and hi > lo and 0 < lo - self.w_lo <= tail_params:
                    segs.append((cur, lo, hi, jobs))
                    cur, jobs, hi = [], [], lo
                    tail_params = 0
        if cur:
            segs.append((cur, self.w_lo, hi, jobs))
        on_gpu = torch.device(net.dev).type == "cuda"
        groups = net.make_wgrad_groups([j for _, _, _, j in segs]) if on_gpu else []
        self.wgrad_groups = groups
        self._seg_groups, gi = [], 0
        for _, _, _, jobs in segs:
            if jobs and on_gpu:
                self._seg_groups.append(groups[gi])
                gi += 1
            else:
                self._seg_groups.append(None)
        return [(fns, lo, hi) for fns, lo, hi, _ in segs]

    def _run_segment(self, i):
        """Segment i in stream order: its backward chain, then its grouped weight gradient on every CU."""
        for f in self._segments[i][0]:
            f()
        if self._seg_groups[i] is not None:
            self._seg_groups[i].launch()

    # ---------------------------------------------------------------------------- step
    def _front(self):
        net = self.net
        # The step's clearing launch (gradient buffer, BN-backward accumulators, control block: 0.23 GB) touches nothing the forward
        # pass or the loss reads or writes.  With the fused convolution + BN-apply launches (MBX_FUSE_APPLY=1) it comes FIRST: their
        # grid barriers may raise the step control word in the forward pass, which this launch clears -- the step is then skipped
        # like one whose backward barrier timed out.  Otherwise behind the loss, where it measured 38 us against 56 us at the top
        # of the step (behind the optimiser's 1.8 GB of writes).  (Beside the forward pass on a stream of its own: 0.4 ms SLOWER,
        # a fork / join in the captured graph costs more than it hides -- LAB_NOTES round 5; removed.)
        first = net.fuse_apply
        if first:
            net.zero_grads()
        net.set_input(self.images)
        net.forward()
        self.loss.forward_backward(net.locs, net.logits, self.gt, self.n_gt)
        if not first:
            net.zero_grads()

    def _capture(self):
        """Warm up eagerly once (lazy module loads, hipFuncSetAttribute), then capture.  The warm-up pass is not a training
        step: the BN moving statistics it touched are put back (the captured step that follows would otherwise apply this
        batch's moving-average update twice -- at the first step and after every in-process re-capture)."""
        net = self.net
        self._front_merged = False
        mm, mv = net.MM.clone(), net.MV.clone()
        self._front()
        for i in range(len(self._segments)):
            self._run_segment(i)
        net.MM.copy_(mm)
        net.MV.copy_(mv)
        torch.cuda.synchronize()
        graphs = []
        nseg, K = len(self._segments), self.overlap_cus
        mode = dict(capture_error_mode="thread_local")     # RCCL's watchdog thread may query events while we capture
        if K and self._side is None:
            self._side = torch.cuda.Stream()

        def chain(i, capped):
            net.cu_cap = net.chain_cap if capped else 0      # persistent chain launches leave K CUs to the weight gradient
            try:
                for f in self._segments[i][0]:
                    f()
            finally:
                net.cu_cap = 0

        def beside(fn):
            # fork: `fn` on the side stream, ordered behind everything enqueued so far on the capturing stream (and behind
            # the side stream's earlier work); the caller joins with main.wait_stream(self._side)
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                fn()
            forked[0] = True

        def join():
            if forked[0]:
                torch.cuda.current_stream().wait_stream(self._side)
                forked[0] = False
        forked = [False]
        if K and not self.reducer.enabled:
            # ONE graph.  The weight-gradient groups form their own pipeline on the side stream: group i starts when the
            # chain of segment i is done (and group i-1 is), K persistent workgroups each; the chain never waits for it.
            # The last group has nothing left to run beside: joined first, every CU.
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **mode):
                self._front()
                for i in range(nseg):
                    chain(i, capped=i > 0)
                    grp = self._seg_groups[i]
                    if grp is None:
                        continue
                    if i == nseg - 1:
                        join()
                        grp.launch()
                    else:
                        beside(lambda grp=grp: grp.launch(K))
                join()
            self.graphs = [g]
            return
        if not K and not self.reducer.enabled and nseg == 1 and os.environ.get("MBX_ONE_GRAPH", "1") != "0":
            # single GPU: forward + loss + the whole backward pass + the one grouped weight gradient in ONE graph
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **mode):
                self._front()
                self._run_segment(0)
            self.graphs = [g]
            return
        # data-parallel form without overlapped weight gradients: graph 0 = forward + loss + segment 0, graph i = segment i --
        # one graph per gradient bucket (round 4 replayed the forward pass as a graph of its own: one more launch gap)
        self._front_merged = not K
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, **mode):
            self._front()
            if self._front_merged:
                self._run_segment(0)
        graphs.append(g)
        if K:
            # data-parallel form: graph i = chain of segment i BESIDE the weight gradients of segment i-1 (joined at its end,
            # so that bucket i-1 can be handed to RCCL between two graphs); one more graph for the last group
            for i in range(nseg + 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **mode):
                    prev = self._seg_groups[i - 1] if i > 0 else None
                    if i == nseg:
                        if prev is not None:
                            prev.launch()
                    else:
                        if prev is not None:
                            beside(lambda prev=prev: prev.launch(K))
                        chain(i, capped=prev is not None)
                        join()
                graphs.append(g)
        else:
            for i in range(1, nseg):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **mode):
                    self._run_segment(i)
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
        reduce = self._reduce_bucket
        if self.use_graph:
            if self.graphs is None:
                self._capture()
            self.graphs[0].replay()
            if self.overlap_cus and red.enabled:
                # graph i + 1 = chain i beside the weight gradients of segment i - 1: bucket i - 1 is complete behind it
                for i, g in enumerate(self.graphs[1:]):
                    g.replay()
                    if i > 0:
                        reduce(i - 1, *self._segments[i - 1][1:])
            elif getattr(self, "_front_merged", False) and len(self.graphs) == len(self._segments):
                reduce(0, *self._segments[0][1:])              # (graph 0 held segment 0)
                for i, (g, (_, lo, hi)) in enumerate(zip(self.graphs[1:], self._segments[1:]), start=1):
                    g.replay()
                    reduce(i, lo, hi)
            else:
                for i, (g, (_, lo, hi)) in enumerate(zip(self.graphs[1:], self._segments)):
                    g.replay()
                    reduce(i, lo, hi)
        else:
            self._front()
            for i, (_, lo, hi) in enumerate(self._segments):
                self._run_segment(i)
                reduce(i, lo, hi)
        if self.exposed_events is not None and red.enabled:
            # diagnostics (bench.py, N > 1): how long the training stream sits behind the last collectives -- the part of the
            # all-reduce the backward pass did not hide
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            red.wait()
            e1.record()
            self.exposed_events.append((e0, e1))
        else:
            red.wait()
        self._optimizer()
        self.global_step += 1

    def _reduce_bucket(self, i, lo, hi):
        """Start the all-reduce of bucket i = Wg[lo:hi].  The LAST bucket (lowest addresses: the bottom of the network)
        takes the beta gradients and the step control block, which lie right below Wg in net.G, along in the same
        collective -- unless frozen filter gradients lie between (--fine_tune: w_lo > 0), then they go separately."""
        net, red = self.net, self.reducer
        if not red.enabled:
            return
        if i < len(self._segments) - 1:
            red.reduce_async(net.Wg, lo, hi)
            return
        net.step_ctl[1:2].copy_(self._stop_flag)       # (G was zeroed inside the step: control word [1] = this rank's request)
        if lo == 0:
            red.reduce_async(net.G, self.bt_lo, net.G_off + hi)
        else:
            red.reduce_async(net.Wg, lo, hi)
            red.reduce_async(net.Btg, self.bt_lo, net.nBt + 8)

    def run_eager_once(self):
        """forward + loss + backward launched eagerly (profiling aid; no optimizer, no all-reduce)."""
        self._front()
        for i in range(len(self._segments)):
            self._run_segment(i)

    def _optimizer(self):
        net, l = self.net, _lib.lib()
        s = torch.cuda.current_stream().cuda_stream
        t = self.global_step
        lr = learning_rate(t, self.lr0, self.dsteps, self.lr_factor, self.staircase)
        d = min(self.ema_decay, (1.0 + t) / (10.0 + t))        # ExponentialMovingAverage(num_updates=global_step)
        self.lr = lr
        # (net.reg_loss was cleared by the step's mbx_step_begin launch: Net.zero_grads)
        P = lambda t_, off=0: None if t_ is None else t_.data_ptr() + 4 * off
        ctl = net.step_ctl.data_ptr()                  # non-zero -> every launch below is a no-op (poisoned step / stop request)
        for buf, lo, hi, on in self.opt_ranges:
            n = hi - lo
            if n <= 0:
                continue
            if buf == "W":
                _lib.check(l.mbx_rmsprop_ema_step(P(net.W, lo), P(net.Wg, lo), P(self.Wms, lo), P(self.Wmom, lo), P(self.Wema, lo),
                                                  net.Wb.data_ptr() + 2 * lo, n, lr, self.rms_decay, self.momentum, self.eps,
                                                  WEIGHT_DECAY, d, on, net.reg_loss.data_ptr(), ctl, s), "rmsprop W")
            else:
                _lib.check(l.mbx_rmsprop_ema_step(P(net.Bt, lo), P(net.Btg, lo), P(self.Btms, lo), P(self.Btmom, lo), P(self.Btema, lo),
                                                  None, n, lr, self.rms_decay, self.momentum, self.eps, 0.0, d, on, None, ctl, s), "rmsprop beta")
        # moving statistics <- this step's batch statistics, and their EMA shadows (train.py:257) from the new values: one gated launch
        assert self.bt_lo == (net.head_bt_start if net.fine_tune else 0)
        net.apply_moving_update(net.step_ctl, self.skipped_steps, self.MMema, self.MVema, d)
        net.prepare_filters()

    def losses(self):
        """(location_loss, confidence_loss, regularization_loss, total_loss) -- host sync."""
        l2 = self.loss.loss2.tolist()
        reg = float(self.net.reg_loss) + self.frozen_reg
        return l2[0], l2[1], reg, l2[0] + l2[1] + reg

    def match_status(self):
        return self.loss.status
