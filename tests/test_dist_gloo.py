"""world_size-2 gloo test of the data-parallel gradient path (runs on CPU): the bucketed async
all-reduce over the backward segments sums every trainable gradient element exactly once."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, fine_tune, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    from multibox_amd.dist import BucketReducer, shard_range
    net = Net(batch=1, input_size=299, k=5, mode="train", fine_tune=fine_tune, device="cpu")
    tr = Trainer.__new__(Trainer)
    tr.net, tr.w_lo, tr.bt_lo = net, (net.head_w_start if fine_tune else 0), (net.head_bt_start if fine_tune else 0)
    # the product's own bucket path (Trainer.step): four buckets + the small tail bucket of data-parallel runs, the last
    # collective taking the beta gradients and the step control block along where they are contiguous with it
    tr._segments = segs = tr._make_segments(4, tail_params=2_000_000)
    gen = torch.Generator().manual_seed(rank)
    net.Wg.copy_(torch.randn(net.nW, generator=gen))
    net.Btg.copy_(torch.randn(net.nBt + 8, generator=gen))     # beta gradients + the step control block (engine.py)
    tr._stop_flag = net.Btg[net.nBt + 1].clone()               # (step() copies the rank's stop request into control word 1)
    mine_w, mine_b = net.Wg.clone(), net.Btg.clone()
    red = tr.reducer = BucketReducer(dist.group.WORLD)
    assert red.enabled
    n_coll = 0
    for i, (_, lo, hi) in enumerate(segs):
        tr._reduce_bucket(i, lo, hi)
    n_coll = len(red.works)
    red.wait()
    assert n_coll == len(segs) + (1 if fine_tune else 0), (n_coll, len(segs))     # full training: no separate collective for Btg
    other = torch.Generator().manual_seed(1 - rank)
    ow = torch.randn(net.nW, generator=other)
    ob = torch.randn(net.nBt + 8, generator=other)
    ok = torch.allclose(net.Wg[tr.w_lo:], (mine_w + ow)[tr.w_lo:]) and torch.allclose(net.Btg[tr.bt_lo:], (mine_b + ob)[tr.bt_lo:])
    ok = ok and torch.equal(net.Wg[:tr.w_lo], mine_w[:tr.w_lo])            # frozen range untouched
    ok = ok and shard_range(10, rank, world) == ((0, 5) if rank == 0 else (5, 10))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("fine_tune", [False, True])
def test_bucketed_allreduce_world2(fine_tune):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (1 if fine_tune else 0)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, fine_tune, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _detect_worker(rank, world, port, q):
    """Rank-sharded detect: disjoint batches per rank, rank 0 concatenates in single-process order."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multibox_amd import detect as D
    batches = [dict(ids=list(range(4 * i, 4 * i + 4))) for i in range(7)]          # 7 batches of 4 patches
    local = []
    for bi, b in D.shard_batches(iter(batches), rank, world):
        local.append((bi, [{"image_id": i, "bbox": [0.0, 0.0, 1.0, 1.0], "score": 1.0 / (1 + i)} for i in b["ids"]]))
    merged = D.gather_results(local)
    if rank == 0:
        single = D.merge_results([[(bi, [{"image_id": i, "bbox": [0.0, 0.0, 1.0, 1.0], "score": 1.0 / (1 + i)} for i in b["ids"]])
                                   for bi, b in enumerate(batches)]])
        ok = merged == single and [r["image_id"] for r in merged] == list(range(28))
    else:
        ok = merged is None
    # ---- the INPUT is sharded too: from real JPEG records, each rank decodes only the records that feed
    # its own batches; the merged result text is byte-identical to the single-process file
    import json
    import numpy as np
    from multibox_amd import inputs as I, records as REC
    from multibox_amd.config import Cfg
    path = os.path.join(os.environ["MBX_TEST_TMP"], "shard.tfrecords")
    cfg = Cfg(dict(INPUT_SIZE=64, DETECTION=dict(USE_ORIGINAL_IMAGE=True, ORIGINAL_IMAGE_MAX_TO_KEEP=5, USE_FLIPPED_ORIGINAL_IMAGE=True,
                   FLIPPED_IMAGE_MAX_TO_KEEP=5, CROPS=[dict(HEIGHT=48, WIDTH=48, HEIGHT_STRIDE=40, WIDTH_STRIDE=40, FLIP=False, MAX_TO_KEEP=3)])))

    def fake_results(b):
        """Deterministic 'detections' of a batch from its own pixels and metadata (stands in for the GPU forward)."""
        B = len(b["image_ids"])
        px = np.stack([b["images"][i][:2, :2, 0].reshape(-1) for i in range(B)]).astype(np.float64)      # [B, 4]
        boxes = np.repeat(px[:, None, :], 5, 1) + b["offsets"][:, None, :1] + np.arange(5)[None, :, None] * 0.125
        scores = (boxes[:, :, 0] * 0.01).astype(np.float32)
        count = np.minimum(b["max_to_keep"].reshape(-1), 5).astype(np.int32)
        ids = [int(i) for i in b["image_ids"]]
        return REC.batch_chunk(boxes, scores, count, ids)
    st = {}
    mine = [(b["batch_index"], [fake_results(b)[1]]) for b in I.detect_batches([path], cfg, 4, keep_partial=True, rank=rank, world=world, stats=st)]
    merged2 = D.gather_results(mine)
    if rank == 0:
        st1 = {}
        single2 = [fake_results(b)[1] for b in I.detect_batches([path], cfg, 4, keep_partial=True, stats=st1)]
        text_single = REC.records_to_json(single2)
        ok = ok and REC.records_to_json(merged2) == text_single and len(json.loads(text_single)) > 20
        ok = ok and st1 == {"records": 6, "decoded": 6}
    ok = ok and st["records"] == 6 and 0 < st["decoded"] < 6          # this rank skipped whole records without decoding them
    q.put((rank, bool(ok), len(local), st["decoded"]))
    dist.destroy_process_group()


def test_detect_sharding_world2(tmp_path):
    import torch.multiprocessing as mp
    from tests.test_inputs_cpu import _make_records
    # six pictures of different sizes: 2 + (number of 48x48 windows at stride 40) patches each, batches of 4
    _make_records(str(tmp_path / "shard.tfrecords"), [(64, 64, []), (100, 140, []), (90, 90, []), (64, 200, []), (130, 64, []), (70, 75, [])])
    os.environ["MBX_TEST_TMP"] = str(tmp_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_detect_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
    assert [r[:3] for r in res] == [(0, True, 4), (1, True, 3)], res
    assert all(0 < r[3] < 6 for r in res), res                      # decode counts: nobody decoded every record


def _bcast_worker(rank, world, port, q):
    """Trainer.broadcast_parameters: every rank ends with rank 0's variables / slots / shadows / step;
    Trainer.check_health: a failure flag on ONE rank raises on EVERY rank."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    net = Net(batch=1, input_size=299, k=5, mode="train", device="cpu", seed=2 + rank)      # DIFFERENT weights per rank
    tr = Trainer.__new__(Trainer)
    tr.net, tr.pg, tr.w_lo, tr.bt_lo, tr.global_step = net, dist.group.WORLD, 0, 0, 10 * (rank + 1)
    f = lambda n, v: torch.full((n,), float(v))
    tr.Wms, tr.Btms, tr.Wmom, tr.Btmom = f(net.nW, rank + 1), f(net.nBt, rank + 1), None, None
    tr.Wema, tr.Btema, tr.MMema, tr.MVema = net.W.clone(), net.Bt.clone(), net.MM.clone(), net.MV.clone()
    w_before = net.W.clone()
    tr.broadcast_parameters(src=0)
    ref = Net(batch=1, input_size=299, k=5, mode="train", device="cpu", seed=2)
    ok = torch.equal(net.W, ref.W) and torch.equal(tr.Wema, ref.W) and float(tr.Wms[0]) == 1.0 and tr.global_step == 10
    ok = ok and (rank == 0 or not torch.equal(w_before, ref.W))
    tr._timeouts_seen, tr._stop_flag, tr.events, tr.wgrad_groups = 0, torch.zeros(()), [], []
    net.no_onepass = True                          # barrier_timeouts() -> 0 on CPU
    # health: a stop request on ONE rank (its input is exhausted) is reported on EVERY rank, without raising

    class OkLoss:
        status = torch.zeros(2, dtype=torch.int32)
    tr.loss = OkLoss()
    ok = ok and tr.check_health() == {"stop": False, "fallback": False}
    if rank == 1:
        tr.request_stop()
    ok = ok and tr.check_health()["stop"] is True
    # ... and it is evaluated FIRST: with the request up, a failed matching on the exhausted rank (its repeated batch;
    # those steps were not applied) ends the run cleanly instead of raising

    class StopLoss:
        status = torch.tensor([0, 2 if rank == 1 else 0], dtype=torch.int32)
    tr.loss = StopLoss()
    ok = ok and tr.check_health() == {"stop": True, "fallback": False}
    tr._stop_flag.zero_()
    # health: rank 1 reports a failed matching, both ranks must raise

    class FakeLoss:
        status = torch.tensor([0, 2 if rank == 1 else 0], dtype=torch.int32)
    tr.loss = FakeLoss()
    try:
        tr.check_health()
        raised = False
    except RuntimeError as e:
        raised = "matching failed" in str(e)
    q.put((rank, bool(ok), raised))
    dist.destroy_process_group()


def test_broadcast_and_health_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == [(0, True, True), (1, True, True)]
