"""Multi-process tests of the data-parallel host logic on CPU (gloo, world_size 2): the same GradAllReduce / shard_batch
code runs over RCCL on the GPUs.  No GPU and no HIP library involved."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import pivp_amd
        dp = pivp_amd.GradAllReduce(nbuckets=4)
        assert dp.world_size == world and dp.rank == rank
        n = 100003                                   # not a multiple of the bucket alignment
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        dp.allreduce_flat(g)
        expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok_sum = bool(torch.equal(g, expect))
        bounds = dp.bucket_bounds(n)
        ok_bounds = bounds[0][0] == 0 and bounds[-1][1] == n and all(a[1] == b[0] for a, b in zip(bounds[:-1], bounds[1:])) \
            and all(a % 64 == 0 for a, _ in bounds)
        # sharding: contiguous batch shards that tile the global batch
        imgs = np.arange(3 * 8 * 2, dtype=np.float32).reshape(3, 8, 2)
        (mine,) = pivp_amd.shard_batch([imgs], rank, world)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ok_shard = np.array_equal(np.concatenate(gathered, axis=1), imgs)
        # data-parallel Adam: SUM all-reduce + gscale = 1/world equals the gradient of the global-batch mean loss
        local_grad = torch.full((10,), float(rank + 1))
        dp.allreduce_flat(local_grad)
        ok_mean = bool(torch.allclose(local_grad / world, torch.full((10,), sum(r + 1 for r in range(world)) / world)))
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, ok_sum, ok_bounds, ok_shard, ok_mean, float(t.item())))
    finally:
        dist.destroy_process_group()


def test_grad_allreduce_and_sharding_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_sum, ok_bounds, ok_shard, ok_mean, tmax in res:
        assert ok_sum and ok_bounds and ok_shard and ok_mean, (rank, ok_sum, ok_bounds, ok_shard, ok_mean)
        assert tmax == float(world)


def test_shard_batch_rejects_ragged():
    sys.path.insert(0, ROOT)
    import pivp_amd
    with pytest.raises(ValueError):
        pivp_amd.shard_batch([np.zeros((3, 7, 2))], 0, 2)
