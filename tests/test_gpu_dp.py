"""Data-parallel training step on the GPU kernels with a REAL world size of 2: two processes share the one GPU of
the test box and exchange gradients through gloo (RCCL refuses two ranks on one device; the reducer is backend-
agnostic).  Checks SURVEY 8(e): the gradient all-reduce is a SUM (the reference loss is a batch sum, loss.py:100-101),
every rank ends the step with the same weights, and those weights equal a single-process step on the summed
gradients of the two shards."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(torch, seed, pg=None, use_graph=True):
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    from multibox_amd import priors as PR
    from multibox_amd.synth import DEFAULT_ASPECT_RATIOS
    pri = np.array(PR.generate_priors(DEFAULT_ASPECT_RATIOS[5]), np.float32)
    # At batch 4 the random-init network is chaotic in the backward pass: the summation order of ONE batch-norm
    # reduction (the one-launch BN backward sums its per-channel totals with fp32 atomics) changes the stem gradients
    # by tens of percent from run to run.  This test therefore runs the deterministic three-launch BN backward in the
    # ranks and in the reference; the weight gradient's atomics are the one remaining noise and do not feed back.
    # (the same holds for the measured tile choice: another tile height regroups the BN partial sums)
    # MBX_DETERMINISTIC=1 also forbids pixel splits in the grouped weight gradient (one adder per dw element): the
    # ranks cut the backward pass into 3 segments and the reference into 1, so the split plans would differ otherwise.
    # (put back afterwards: this also runs in the pytest process itself -- left set, every later test module ran
    # deterministic and un-tuned without saying so, which is how the fused-launch tests of round 6 found it)
    old = {k_: os.environ.get(k_) for k_ in ("MBX_DETERMINISTIC", "MBX_AUTOTUNE")}
    os.environ["MBX_DETERMINISTIC"] = "1"
    os.environ["MBX_AUTOTUNE"] = "0"
    try:
        net = Net(batch=4, input_size=299, k=5, mode="train", seed=seed)
        tr = Trainer(net, pri, max_num_bboxes=13, use_graph=use_graph, process_group=pg)
    finally:
        for k_, v_ in old.items():
            if v_ is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v_
    return net, tr


def _batch(torch, rank, B=4):
    from multibox_amd.synth import synthetic_batch
    images, gt, n = synthetic_batch(B, 299, 13, seed=40 + rank)
    return torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    net, tr = _build(torch, seed=13, pg=dist.group.WORLD, use_graph=True)
    assert tr.reducer.enabled and len(tr._segments) == 3      # two equal buckets, the last one cut once more (small tail): round 5's default
    tr.set_batch(*_batch(torch, rank))
    tr.step()
    torch.cuda.synchronize()
    torch.save({"W": net.W.cpu(), "Bt": net.Bt.cpu(), "Wg": net.Wg.cpu(), "loss": tr.losses(),
                "status": int(tr.match_status().max())}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_summed_gradient_step(tmp_path):
    import torch
    import torch.multiprocessing as mp
    import __graft_entry__ as g
    g.build()
    ctx = mp.get_context("spawn")
    port = 29700 + (os.getpid() % 1000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(600)
        assert p.exitcode == 0
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["status"] == 0 and r1["status"] == 0
    # every rank holds the same summed gradient and the same updated weights (bit for bit: same sums, same optimizer)
    assert torch.equal(r0["Wg"], r1["Wg"]) and torch.equal(r0["W"], r1["W"]) and torch.equal(r0["Bt"], r1["Bt"])
    assert r0["loss"][:2] != r1["loss"][:2]                       # but each saw its own shard
    # single-process reference: gradients of the two shards computed one after the other, summed, one optimizer step
    net, tr = _build(torch, seed=13, pg=None, use_graph=False)
    w0 = net.W.clone()
    grads = []
    mm0, mv0 = net.MM.clone(), net.MV.clone()
    for rank in range(2):
        net.MM.copy_(mm0); net.MV.copy_(mv0)
        tr.set_batch(*_batch(torch, rank))
        tr.run_eager_once()
        torch.cuda.synchronize()
        grads.append((net.Wg.clone(), net.Btg.clone()))
    net.Wg.copy_(grads[0][0] + grads[1][0])
    net.Btg.copy_(grads[0][1] + grads[1][1])
    tr._optimizer()
    torch.cuda.synchronize()

    def cos(a, b):
        a, b = a.double().reshape(-1), b.double().reshape(-1)
        return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    g_ref = (grads[0][0] + grads[1][0]).cpu()
    # deterministic mode: each shard's gradient is bit-reproducible, a two-term sum commutes -> identical, not close
    assert cos(r0["Wg"], g_ref) > 0.999999, cos(r0["Wg"], g_ref)
    assert torch.equal(r0["Wg"], g_ref), float((r0["Wg"] - g_ref).abs().max())
    assert torch.equal(r0["W"], net.W.cpu()) and torch.equal(r0["Bt"], net.Bt.cpu())
    assert not torch.equal(r0["W"], w0.cpu())


def _worker_shipped(rank, world, port, out_dir, B=4):
    """The SHIPPED data-parallel configuration: one-launch batch-norm backward with a capped grid, measured tile table,
    hipGraph segments, bucketed async all-reduce.  Two ranks share this box's one GPU, so each caps its grid-barrier
    kernels at 96 workgroups (two concurrent 192-workgroup grids cannot both be resident on 256 CUs; on the 8-GPU
    node each rank owns a GPU and runs 192)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.pop("MBX_NO_BN_ONEPASS", None)
    os.environ.pop("MBX_DETERMINISTIC", None)
    os.environ.pop("MBX_AUTOTUNE", None)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    from multibox_amd import priors as PR
    from multibox_amd.synth import DEFAULT_ASPECT_RATIOS
    pri = np.array(PR.generate_priors(DEFAULT_ASPECT_RATIOS[5]), np.float32)
    net = Net(batch=B, input_size=299, k=5, mode="train", seed=13 + rank, bn_max_workgroups=96)   # different seeds:
    tr = Trainer(net, pri, max_num_bboxes=13, use_graph=True, process_group=dist.group.WORLD)      # broadcast fixes it
    tr.broadcast_parameters(src=0)
    n_onepass = sum(1 for op in net.convs if getattr(op, "bn_ws_off", -1) >= 0)
    n_persistent = sum(1 for _, d, _ in net.tune_registry if d.tile_config > 32)
    tr.set_batch(*_batch(torch, rank, B))
    healthy, fallback, checks = True, False, []
    for rnd in range(3):                     # steps, health check (may fall back, both ranks together), more steps
        n_steps = 3 if B == 4 else 2         # (B = 64: 240 MB of gradients cross gloo's host path every step)
        before = net.barrier_timeouts()
        for _ in range(n_steps):
            tr.step()
        torch.cuda.synchronize()
        rec = {"round": rnd, "steps": n_steps, "bn_backward_before": "three-launch" if net.no_onepass else "one-launch",
               "timeouts_in_round": net.barrier_timeouts() - before, "error": None}
        try:
            h = tr.check_health()
            fallback = h["fallback"] or fallback
            rec.update(fallback_taken=bool(h["fallback"]), skipped_steps_total=int(tr.skipped_steps), global_step=tr.global_step)
        except RuntimeError as e:
            healthy = False
            rec["error"] = str(e)
        checks.append(rec)
    torch.save({"W": net.W.cpu(), "Bt": net.Bt.cpu(), "Wg": net.Wg.cpu(), "loss": tr.losses(), "healthy": healthy,
                "fallback": fallback, "timeouts": net.barrier_timeouts(), "onepass_layers": n_onepass,
                "persistent_launches": n_persistent, "events": tr.events, "checks": checks,
                "bn_backward_at_end": "three-launch" if net.no_onepass else "one-launch",
                "applied_steps": tr.global_step, "skipped_steps": int(tr.skipped_steps)},
               os.path.join(out_dir, "s_rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 64])
def test_two_rank_shipped_configuration(tmp_path, B):
    """B = 64 is BASELINE config 3's per-GPU batch: the shipped tile table sends ~175 launches per step to the persistent
    igemm5 kernel (queued tiles), and the two ranks' persistent kernels and capped grid-barrier kernels contend for the
    ONE GPU of this box -- far heavier sharing than RCCL's kernels cause on a rank's own GPU.  A grid-barrier time-out
    under that load is allowed if (and only if) the trainer handled it: the step skipped on both ranks, the fall-back taken
    together, training continued, parameters still identical."""
    import torch
    import torch.multiprocessing as mp
    import __graft_entry__ as g
    g.build()
    ctx = mp.get_context("spawn")
    port = 30700 + (os.getpid() % 1000) + B
    ps = [ctx.Process(target=_worker_shipped, args=(r, 2, port, str(tmp_path), B)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(900)
        assert p.exitcode == 0
    r0, r1 = torch.load(tmp_path / "s_rank0.pt"), torch.load(tmp_path / "s_rank1.pt")
    # WHICH branch happened is part of the result: written to gpurun_out/ (merged back from the GPU box)
    import json
    branch = {"batch_per_rank": B, "bn_max_workgroups": 96, "ranks_on_one_gpu": 2,
              "branch": "fallback to the three-launch BN backward" if r0["fallback"] else "one-launch BN backward held",
              "rank0": {k: r0[k] for k in ("checks", "timeouts", "fallback", "bn_backward_at_end", "applied_steps", "skipped_steps",
                                           "onepass_layers", "persistent_launches", "healthy")},
              "rank1": {k: r1[k] for k in ("checks", "timeouts", "fallback", "bn_backward_at_end", "applied_steps", "skipped_steps", "healthy")}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "test_two_rank_shipped_B%d.json" % B), "w") as f:
        json.dump(branch, f, indent=1)
    assert r0["onepass_layers"] > 100                                   # the one-launch BN backward really ran
    # a time-out AFTER the fall-back (or any other failed health check) fails the test on either rank
    assert r0["healthy"] and r1["healthy"], branch
    assert r0["applied_steps"] == r1["applied_steps"] and r0["skipped_steps"] == r1["skipped_steps"], branch   # skipped together
    if B == 4:
        assert r0["timeouts"] == 0 and r1["timeouts"] == 0 and not r0["fallback"], branch
    else:
        assert r0["persistent_launches"] > 100                          # queued igemm5 launches under contention
        assert r0["fallback"] == r1["fallback"], branch
        if r0["fallback"]:
            # taken once, together; every round AFTER it clean (no time-outs, no skipped steps) and on the three-launch form
            after = [c for c in r0["checks"] + r1["checks"] if c["bn_backward_before"] == "three-launch"]
            assert after and all(c["timeouts_in_round"] == 0 and c["error"] is None for c in after), branch
            assert r0["bn_backward_at_end"] == r1["bn_backward_at_end"] == "three-launch", branch
        else:
            assert r0["timeouts"] + r1["timeouts"] == 0 and r0["skipped_steps"] == 0, branch
    # four / six steps on: all-reduced gradients and weights identical on both ranks, bit for bit
    assert torch.equal(r0["Wg"], r1["Wg"]) and torch.equal(r0["W"], r1["W"]) and torch.equal(r0["Bt"], r1["Bt"])
    assert bool(torch.isfinite(r0["W"]).all()) and all(np.isfinite(x) for x in r0["loss"])
    assert r0["loss"][:2] != r1["loss"][:2]
