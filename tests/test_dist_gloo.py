"""CPU: the N>1 path (row sharding + gather of logL to rank 0) with world_size 2 and 3 over gloo.
The evaluator is a stand-in (a deterministic function of the row), because the HIP path needs a
GPU; what is under test is exactly the code bench.py and a multi-GPU caller use: shard_bounds,
LogLGather and sharded_loglike."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mcalf_amd  # noqa: F401
from mcalf_amd import dist as mdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_logl(P):
    return (P * np.arange(1, P.shape[1] + 1)).sum(axis=1) - 0.5 * (P ** 2).sum(axis=1)


def _worker(rank, world, port, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = np.random.default_rng(5).random((batch, 7))      # same matrix on every rank

        def evaluate(lo, hi, out):
            out.copy_(torch.from_numpy(_fake_logl(P[lo:hi])))

        full = mdist.sharded_loglike(evaluate, batch, "cpu")
        # a reusable plan gives the same answer twice (buffers are not stale)
        plan = mdist.LogLGather(batch, "cpu")
        res = []
        for _ in range(2):
            evaluate(plan.lo, plan.hi, plan.local)
            res.append(plan.gather())
        # pipelined form: two buffers in flight, gather of batch k overlaps the evaluation of k+1
        ring = mdist.LogLGather(batch, "cpu", depth=2)
        for k in range(5):
            out = ring.local
            out.copy_(torch.from_numpy(_fake_logl(P[ring.lo:ring.hi]) + k))
            ring.gather_async()
        last = ring.finish()
        if rank == 0:
            assert np.array_equal(last.numpy(), _fake_logl(P) + 4)
            q.put((full.numpy().copy(), res[0].numpy().copy(), res[1].numpy().copy()))
        else:
            assert full is None and res[0] is None and last is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 64), (2, 7), (3, 10), (2, 1)])
def test_sharded_equals_unsharded(world, batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = _fake_logl(np.random.default_rng(5).random((batch, 7)))
    for g in got:
        assert np.array_equal(g, want)          # bit for bit, ragged shards included


def test_shard_bounds_cover_and_balance():
    for batch in (0, 1, 5, 1024, 32768, 1000003):
        for world in (1, 2, 3, 8):
            spans = [mdist.shard_bounds(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
            assert sizes == mdist.shard_counts(batch, world)
    with pytest.raises(ValueError):
        mdist.shard_bounds(10, 2, 2)


def test_single_process_plan_is_identity():
    plan = mdist.LogLGather(5, "cpu")
    plan.local.copy_(torch.arange(5, dtype=torch.float64))
    assert torch.equal(plan.gather(), torch.arange(5, dtype=torch.float64))


def _one_rank_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        ring = mdist.LogLGather(6, "cpu", depth=2, always_collective=True)
        assert ring.collective
        for k in range(3):
            ring.local.copy_(torch.arange(6, dtype=torch.float64) + k)
            ring.gather_async()
        q.put(ring.finish().numpy().copy())
    finally:
        dist.destroy_process_group()


def test_one_rank_can_still_run_the_collective():
    """bench.py's single-rank rehearsal of the gather path (MCALF_BENCH_FORCE_DIST) relies on this."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=120)
    p.join(timeout=120)
    assert p.exitcode == 0 and np.array_equal(got, np.arange(6.0) + 2)
