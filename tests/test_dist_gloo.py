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
    segs = tr._make_segments(4)
    gen = torch.Generator().manual_seed(rank)
    net.Wg.copy_(torch.randn(net.nW, generator=gen))
    net.Btg.copy_(torch.randn(net.nBt, generator=gen))
    mine_w, mine_b = net.Wg.clone(), net.Btg.clone()
    red = BucketReducer(dist.group.WORLD)
    assert red.enabled
    for _, lo, hi in segs:
        red.reduce_async(net.Wg, lo, hi)
    red.reduce_async(net.Btg, tr.bt_lo, net.nBt)
    red.wait()
    other = torch.Generator().manual_seed(1 - rank)
    ow = torch.randn(net.nW, generator=other)
    ob = torch.randn(net.nBt, generator=other)
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
