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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from host_stub import HostStubModel  # noqa: E402  (the test double of pivp_amd.Model's training protocol)


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


# ---- the overlapped per-group path (GradAllReduce.backward_and_allreduce), two ranks, gloo ------------------------------------

def _overlap_worker(rank, world, port, q, scenario):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import pivp_amd
        dp = pivp_amd.GradAllReduce(payload='fp32' if scenario == 'bf16_model_fp32_payload' else 'auto')
        kw = {}
        if scenario in ('bf16_payload', 'bf16_model_fp32_payload'):
            kw['precision'] = 'bf16'                     # a model in the bf16 precision mode (config 3): 'auto' sends bf16
        if scenario == 'callback_raises' and rank == 1:
            kw['fail_in_group'] = 2                      # rank 1's callback for group 2 raises; rank 0 is healthy
        if scenario == 'group_skipped' and rank == 1:
            kw['skip_groups'] = (3,)                     # rank 1 never announces group 3
        model = HostStubModel(value=float(rank + 1) * (1.003 if 'bf16' in scenario else 1.0), **kw)
        model.cleargrads()
        err = None
        try:
            dp.backward_and_allreduce(model)
        except Exception as e:                           # the failing rank re-raises AFTER matching every collective
            err = '%s: %s' % (type(e).__name__, e)
        flat = model._ensure_grads().clone()
        # a collective issued after the step proves that no rank is still stuck inside the step's collectives
        after = torch.tensor([float(rank + 1)])
        dist.all_reduce(after)
        q.put((rank, err, list(dp.issued), list(model.announced), flat.numpy(), float(after.item()) + 1000.0 * dp.last_payload_bytes))
    finally:
        dist.destroy_process_group()


def _run_overlap(scenario):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q, scenario)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    global _payload_bytes
    _payload_bytes = [int(r[5] // 1000.0) for r in res]
    return [r[:5] + (r[5] % 1000.0,) for r in res]


_payload_bytes = None


def _expected_sum(world=2):
    sys.path.insert(0, ROOT)
    import pivp_amd
    m = HostStubModel()
    out = np.zeros(sum(m.sizes), dtype=np.float32)
    for g, (a, b) in enumerate(m.grad_group_ranges()):
        out[a:b] = (g + 1) * sum(r + 1 for r in range(world))
    return out


def test_overlapped_allreduce_two_ranks():
    """Six gradient groups announced from inside backward(): summed over both ranks, same collective order on both."""
    res = _run_overlap('ok')
    expect = _expected_sum()
    for rank, err, issued, announced, flat, after in res:
        assert err is None
        assert issued == [0, 1, 2, 3, 4, 5] and announced == [0, 1, 2, 3, 4, 5]
        assert np.array_equal(flat, expect)
        assert after == 3.0


def test_overlapped_allreduce_survives_a_failing_callback():
    """One rank's callback raises in group 2: that rank still issues all six collectives in order and re-raises afterwards; the
    healthy rank returns from every collective (nobody blocks), and a later collective still works on both."""
    res = _run_overlap('callback_raises')
    (r0, err0, issued0, _, flat0, after0), (r1, err1, issued1, _, _, after1) = res
    assert err0 is None and issued0 == [0, 1, 2, 3, 4, 5]
    assert err1 is not None and 'injected failure in group 2' in err1
    assert issued1 == [0, 1, 2, 3, 4, 5]                 # same sequence as the healthy rank
    assert after0 == 3.0 and after1 == 3.0
    assert np.array_equal(flat0, _expected_sum())        # the stub's gradients were complete when they were sent


def test_overlapped_allreduce_flags_a_missing_group():
    res = _run_overlap('group_skipped')
    (r0, err0, issued0, _, _, after0), (r1, err1, issued1, _, _, after1) = res
    assert err0 is None and issued0 == [0, 1, 2, 3, 4, 5]
    assert issued1 == [0, 1, 2, 3, 4, 5]
    assert err1 is not None and 'out of order' in err1
    assert after0 == 3.0 and after1 == 3.0


def test_bf16_gradient_payload_two_ranks():
    """BASELINE.json config 3: a model in the bf16 precision mode sends its gradient groups as bf16 (half the bytes); the sum comes
    back into the fp32 flat buffer within bf16 rounding; both ranks hold the same bytes."""
    res = _run_overlap('bf16_payload')
    n = sum(HostStubModel().sizes)
    assert _payload_bytes == [2 * n, 2 * n]
    exact = _expected_sum() * 1.003
    for rank, err, issued, announced, flat, after in res:
        assert err is None and issued == [0, 1, 2, 3, 4, 5]
        rel = np.abs(flat - exact) / exact
        assert rel.max() < 3 * 2.0 ** -8 and rel.max() > 0          # two roundings to bf16 and one bf16 add; NOT the exact fp32 sum
        assert np.array_equal(flat, flat.astype(np.float32).view(np.uint32).__and__(0xFFFF0000).view(np.float32))   # bf16 numbers
    assert np.array_equal(res[0][4], res[1][4])


def test_bf16_model_can_opt_out_of_the_bf16_payload():
    res = _run_overlap('bf16_model_fp32_payload')
    n = sum(HostStubModel().sizes)
    assert _payload_bytes == [4 * n, 4 * n]
    m = HostStubModel()
    exact = np.zeros(n, dtype=np.float32)
    for g, (a, b) in enumerate(m.grad_group_ranges()):
        exact[a:b] = np.float32(np.float32(1.003) * (g + 1)) + np.float32(np.float32(2.006) * (g + 1))
    for rank, err, issued, announced, flat, after in res:
        assert err is None
        assert np.allclose(flat, exact, rtol=1e-6)


# ---- the all-links schedule (all_to_all of shards, local fp32 sum, all_gather) against the plain all-reduce -------------

def _rsag_worker(rank, world, port, q, algo, payload):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import pivp_amd
        dp = pivp_amd.GradAllReduce(payload=payload, algo=algo)
        model = HostStubModel(precision='bf16', random_seed=100 + rank)
        model.cleargrads()
        dp.backward_and_allreduce(model)
        q.put((rank, model._ensure_grads().numpy().copy(), dp.last_algo, list(dp.issued), dp.last_payload_bytes))
    finally:
        dist.destroy_process_group()


def _run_rsag(world, algo, payload):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rsag_worker, args=(r, world, port, q, algo, payload)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _rank_gradients(world):
    from host_stub import local_gradient
    m = HostStubModel()
    out = []
    for r in range(world):
        flat = torch.zeros(sum(m.sizes))
        for g, (a, b) in enumerate(m.grad_group_ranges()):
            flat[a:b] = local_gradient(100 + r, g, b - a)
        out.append(flat)
    return out


@pytest.mark.parametrize('world', [2, 8])
def test_rs_ag_bf16_payload_is_one_rounding_of_the_fp32_sum(world):
    """algo='rs_ag' (the default for a bf16 payload): every element equals bf16(sum in fp32, rank order, of the ranks' bf16-rounded
    gradients) -- ONE rounding of the sum, not one per hop -- and all ranks hold identical bytes.  Group sizes 70 and 1 are padded shards."""
    res = _run_rsag(world, 'auto', 'auto')
    grads = _rank_gradients(world)
    acc = torch.zeros_like(grads[0])
    for gr in grads:                                     # rank order, fp32 accumulate
        acc += gr.bfloat16().float()
    expect = acc.bfloat16().float().numpy()
    n = acc.numel()
    for rank, flat, algo, issued, nbytes in res:
        assert algo == 'rs_ag' and issued == [0, 1, 2, 3, 4, 5] and nbytes == 2 * n
        assert np.array_equal(flat.view(np.uint32), expect.view(np.uint32)), 'rank %d: not bf16(fp32 sum)' % rank
    for r in res[1:]:
        assert np.array_equal(r[1].view(np.uint32), res[0][1].view(np.uint32))


def test_rs_ag_beats_the_bf16_allreduce_at_eight_ranks():
    """The same gradients through algo='allreduce' with the bf16 payload: the reduction itself runs in bf16 (a rounding per hop), so its
    distance from the exact sum is larger than rs_ag's single rounding.  (What VERDICT r03 item 4 / ADVICE r03 asked to be shown.)"""
    world = 8
    grads = _rank_gradients(world)
    exact = torch.stack([g.bfloat16().double() for g in grads]).sum(0).numpy()
    err = {}
    for algo in ('allreduce', 'rs_ag'):
        res = _run_rsag(world, algo, 'bf16')
        assert all(r[2] == algo for r in res)
        for r in res[1:]:
            assert np.array_equal(r[1].view(np.uint32), res[0][1].view(np.uint32))
        err[algo] = float(np.sqrt(np.mean((res[0][1].astype(np.float64) - exact) ** 2)))
    assert err['rs_ag'] < 0.8 * err['allreduce'], err


def test_rs_ag_fp32_payload_matches_the_allreduce():
    """fp32 payload through the all-links schedule: the fp32 sum in rank order."""
    world = 2
    res = _run_rsag(world, 'rs_ag', 'fp32')
    grads = _rank_gradients(world)
    expect = (grads[0] + grads[1]).numpy()
    for rank, flat, algo, issued, nbytes in res:
        assert algo == 'rs_ag' and nbytes == 4 * expect.size
        assert np.array_equal(flat, expect)
