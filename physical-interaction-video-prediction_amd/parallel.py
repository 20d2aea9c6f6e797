"""Data parallelism over the batch axis: one process per GPU, replicated parameters, one sum all-reduce of the flat
gradient buffer per step (the reference itself is single-device: its `optimizer.update`, TM:950, is the
call this wraps).

On MI355X the backend is "nccl" (= RCCL over xGMI); the same code runs with "gloo" on CPU tensors, which is how the
host logic -- including the overlapped, per-group path -- is tested without a GPU (tests/test_parallel_gloo.py)."""
import os

import torch
import torch.distributed as dist


class GradAllReduce(object):
    """Sum-all-reduce of a flat gradient buffer in `nbuckets` contiguous slices on a side stream.

    xGMI is a full mesh of point-to-point links, so a few large slices (not many small ones) keep every link busy;
    the default of 4 buckets of ~9 MB each lets the first slices travel while the tail of backward still runs when the
    caller invokes `allreduce_range` per finished region (the plan fills gradients back-to-front)."""

    def __init__(self, group=None, nbuckets=4, defer_side_wait=True, payload='auto', algo='auto'):
        """payload: what travels in the overlapped all-reduce (`backward_and_allreduce`).  'fp32' = the flat gradient buffer itself
        (36.9 MB per step at 64x64).  'bf16' = every announced group slice is rounded to bf16 into a send buffer on the collective's
        stream (one HIP launch), the bf16 image travels (18.4 MB) and the sum is written back into the fp32 flat buffer,
        which stays the optimizer's input ("bf16 payload + fp32 master accumulation").  'auto' (default) = 'bf16'
        for a model in the bf16 precision mode (BASELINE.json config 3), 'fp32' otherwise; pass 'fp32' to opt out.

        algo: how a group slice is summed over the ranks.
          'allreduce'  one `dist.all_reduce(SUM)` per slice.  With a bf16 payload the REDUCTION ITSELF then runs in bf16 inside RCCL /
                       gloo: the running sum is rounded to 8 significant bits at every hop of the ring (N - 1 roundings, about 2^-9
                       relative each), and "fp32 accumulation" holds only for the buffer the result is written into afterwards.
          'rs_ag'      SURVEY.md 5's all-links schedule: `all_to_all_single` of the slice cut into N shards (rank r receives shard r of
                       every peer, each over its own xGMI link -- the mesh is point-to-point, so all 7 links of a GPU carry 1/N of the
                       slice at once instead of a ring's one neighbour), the N received shards summed locally IN FP32 in rank order and
                       rounded ONCE to the payload type (one HIP launch, `pivp_grad_sum_shards`), then `all_gather_into_tensor` of the
                       reduced shards.  Every shard is reduced by exactly one rank, so all ranks end with identical bytes, and a bf16
                       payload carries one rounding of the exact-in-fp32 sum instead of N - 1.  Same bytes per link as a ring
                       all-reduce: 2 (N - 1) / N of the slice.
          'auto'       (default) 'rs_ag' for a bf16 payload, 'allreduce' for fp32 (RCCL's own fp32 sum has nothing to gain in
                       accuracy); the environment variable PIVP_ALLREDUCE_ALGO overrides 'auto' only."""
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        if payload not in ('auto', 'fp32', 'bf16'):
            raise ValueError("payload must be 'auto', 'fp32' or 'bf16'")
        if algo not in ('auto', 'allreduce', 'rs_ag'):
            raise ValueError("algo must be 'auto', 'allreduce' or 'rs_ag'")
        if algo == 'auto':
            env = os.environ.get('PIVP_ALLREDUCE_ALGO', 'auto')
            if env not in ('auto', 'allreduce', 'rs_ag'):
                raise ValueError("PIVP_ALLREDUCE_ALGO must be 'auto', 'allreduce' or 'rs_ag', got %r" % env)
            algo = env
        self.algo = algo
        self.last_algo = None       # what the last overlapped backward actually ran: 'allreduce' or 'rs_ag'
        self._rsag = {}             # group index -> staging buffers of the rs_ag schedule
        self.payload = payload
        self._send = None           # bf16 send buffer, as large as the flat gradient buffer
        self.last_payload_bytes = 0
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.nbuckets = max(1, int(nbuckets))
        self._stream = None
        self.defer_side_wait = bool(defer_side_wait)   # let only the collective's stream wait for the model's side-stream weight gradients
        self.issued = []          # group indices in the order their collectives were issued by the last overlapped backward

    def _schedule(self, model):
        """-> (bf16 payload?, rs_ag schedule?) for this model under the constructor's `payload` / `algo`"""
        bf16 = self.payload == 'bf16' or (self.payload == 'auto' and getattr(model, 'precision', 'fp32') == 'bf16')
        rs_ag = (self.algo == 'rs_ag') or (self.algo == 'auto' and bf16)
        return bf16, rs_ag

    def _send_buffer(self, flat):
        if self._send is None or self._send.numel() != flat.numel() or self._send.device != flat.device:
            self._send = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
        return self._send

    def _rsag_buffers(self, g, n, bf16, device):
        """stage / recv: N shards of S elements in the payload type (the slice, zero-padded to N * S); red: this rank's reduced shard;
        full: the gathered result.  S is a multiple of 64 elements, so every shard starts 128-B aligned."""
        N = self.world_size
        S = ((n + N - 1) // N + 63) // 64 * 64
        dt = torch.bfloat16 if bf16 else torch.float32
        buf = self._rsag.get(g)
        if buf is None or buf['S'] != S or buf['stage'].dtype != dt or buf['stage'].device != device:
            buf = {'S': S, 'stage': torch.zeros(N * S, dtype=dt, device=device), 'recv': torch.empty(N * S, dtype=dt, device=device),
                   'red': torch.empty(S, dtype=dt, device=device), 'full': torch.empty(N * S, dtype=dt, device=device)}
            self._rsag[g] = buf
        return buf

    def prepare(self, model, probe=True):
        """Everything the overlapped path would otherwise do lazily INSIDE the first backward sweep, on the collective stream's critical path:
        the collective stream, the bf16 send buffer or the rs_ag staging buffers of every gradient group (4 tensors x 6 groups), and -- `probe` --
        one untimed collective of each kind the schedule uses on the payload's dtype, so that RCCL builds its all-reduce or all-to-all /
        all-gather channels here and not in a timed step.  Every rank must call it at the same point (the probes are collectives).
        Returns the kinds probed."""
        flat = model._ensure_grads()
        ranges = model.grad_group_ranges()
        bf16, rs_ag = self._schedule(model)
        cuda = flat.is_cuda
        if cuda and self._stream is None:
            self._stream = torch.cuda.Stream(device=flat.device)
        if rs_ag:
            for g, (a, b) in enumerate(ranges):
                if b > a:
                    self._rsag_buffers(g, b - a, bf16, flat.device)
        elif bf16:
            self._send_buffer(flat)
        probed = []
        if not probe or self.world_size == 1:
            return probed
        N = self.world_size
        dt = torch.bfloat16 if bf16 else torch.float32
        ctx = torch.cuda.stream(self._stream) if cuda else _NullContext()
        if cuda:
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))
        with ctx:
            if rs_ag:
                stage = torch.full((N * 64,), float(self.rank + 1), dtype=dt, device=flat.device)
                recv = torch.empty_like(stage)
                dist.all_to_all_single(recv, stage, group=self.group)
                want = torch.arange(1, N + 1, dtype=torch.float32, device=flat.device).repeat_interleave(64)
                full = torch.empty(N * 64, dtype=dt, device=flat.device)
                dist.all_gather_into_tensor(full, recv[:64].contiguous(), group=self.group)
                ok = bool(torch.equal(recv.float(), want)) and bool((full.float() == 1.0).all())
                probed += ['all_to_all_single', 'all_gather_into_tensor']
            else:
                t = torch.full((64,), 1.0, dtype=dt, device=flat.device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                ok = bool((t.float() == float(N)).all())
                probed += ['all_reduce']
        if cuda:
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)
        if not ok:
            raise RuntimeError('collective probe (%s, %s) returned wrong data on rank %d of %d' % (', '.join(probed), dt, self.rank, N))
        return probed

    def bucket_bounds(self, n):
        """Split [0, n) into nbuckets slices with 64-float aligned edges."""
        edges = [0]
        for i in range(1, self.nbuckets):
            edges.append(min(n, (n * i // self.nbuckets + 63) // 64 * 64))
        edges.append(n)
        return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]

    def allreduce_flat(self, flat):
        """In-place SUM over ranks of a 1-D tensor (device or host)."""
        if self.world_size == 1:
            return flat
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._stream):
                works = [dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                         for a, b in self.bucket_bounds(flat.numel())]
                for w in works:
                    w.wait()
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)
        else:
            for a, b in self.bucket_bounds(flat.numel()):
                dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def allreduce(self, model):
        return self.allreduce_flat(model._ensure_grads())

    def backward_and_allreduce(self, model, force_overlap=False):
        """model.backward() with the all-reduce of each gradient group launched as soon as the sweep has finished that group
        at t = 0 (every parameter is shared by all timesteps, so nothing is final earlier): the heads / enc6 slice travels
        while lstm7 .. enc0 of the last timestep are still being differentiated.  Six contiguous slices of 1-11 MB
        (model.grad_group_ranges()); xGMI is point-to-point, a few large messages keep its links busier than many small ones.

        `model` is anything with `_ensure_grads() -> flat tensor`, `grad_group_ranges() -> [(start, end)]` and
        `backward(on_group=callable)`; device tensors run the collectives on a side stream behind an event, host tensors
        (gloo) issue them asynchronously from the callback.

        Every rank issues the SAME sequence of collectives whatever happens locally: groups go out in ascending order (a group
        announced out of turn first sends the ones before it), and if `backward` raises -- a launch failure, or an exception in a
        callback -- the groups not yet sent are still sent and every collective is waited for before the exception is re-raised.
        Peers therefore leave their collectives (with this rank's unfinished gradients: the step is void) instead of blocking
        in them; the launcher (torch.distributed.run) then ends the job because this rank exits."""
        flat = model._ensure_grads()
        if self.world_size == 1 and not force_overlap:   # force_overlap: exercise the path on one rank (tests)
            model.backward()
            return flat
        ranges = model.grad_group_ranges()
        cuda = flat.is_cuda
        bf16, rs_ag = self._schedule(model)
        self.last_algo = 'rs_ag' if rs_ag else 'allreduce'
        N = self.world_size
        send = self._send_buffer(flat) if (bf16 and not rs_ag) else None
        self.last_payload_bytes = sum(max(0, b - a) for a, b in ranges) * (2 if bf16 else 4)
        if cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            main = torch.cuda.current_stream(flat.device)
        works = []
        self.issued = issued = []
        out_of_turn = []
        # Only the collective's stream has to wait for a group's weight gradients (they run on the model's side stream): with the model's
        # own join switched off the backward sweep's last timestep is not held up at every announcement (0.3 ms per step)
        defer = cuda and self.defer_side_wait and hasattr(model, 'wait_group') and hasattr(model, 'set_group_join')
        if defer:
            model.set_group_join(False)

        def reduce_scatter_gather(g, a, b):
            """the rs_ag schedule for slice [a, b) on the current stream / thread; returns the all-gather's work handle"""
            buf = self._rsag_buffers(g, b - a, bf16, flat.device)
            S = buf['S']
            if cuda and bf16:
                _pack_bf16(flat[a:b], buf['stage'][:b - a], self._stream)
            else:
                buf['stage'][:b - a].copy_(flat[a:b])    # fp32 payload: a copy; host tensors (gloo tests): torch's round-to-nearest-even cast
            # NCCL work.wait() makes the CURRENT stream wait, not the host: the sum below is enqueued behind the exchange
            dist.all_to_all_single(buf['recv'], buf['stage'], group=self.group, async_op=True).wait()
            if cuda:
                _sum_shards(buf['recv'], N, S, buf['red'], self._stream)
            else:
                buf['red'].copy_(buf['recv'].view(N, S).float().sum(0))   # host tensors (gloo tests) only: fp32 sum in rank order, one rounding
            return dist.all_gather_into_tensor(buf['full'], buf['red'], group=self.group, async_op=True)

        def issue(g):
            a, b = ranges[g]
            issued.append(g)
            if b <= a:
                return
            if cuda:
                ev = torch.cuda.Event()
                ev.record(main)                      # everything that writes slice g is already enqueued on `main` ...
                self._stream.wait_event(ev)
                if defer:
                    model.wait_group(g, self._stream)   # ... or on the model's side stream
                with torch.cuda.stream(self._stream):
                    if rs_ag:
                        works.append((g, reduce_scatter_gather(g, a, b)))
                        return
                    if bf16:
                        _pack_bf16(flat[a:b], send[a:b], self._stream)
                    works.append((g, dist.all_reduce(send[a:b] if bf16 else flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
            else:
                if rs_ag:
                    works.append((g, reduce_scatter_gather(g, a, b)))
                    return
                if bf16:
                    send[a:b].copy_(flat[a:b])       # host tensors (gloo tests): torch's round-to-nearest-even cast
                works.append((g, dist.all_reduce(send[a:b] if bf16 else flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

        def on_group(g):
            if g < len(issued):
                out_of_turn.append(g)                # announced twice or late: its collective has already gone out
                return
            if g > len(issued):
                out_of_turn.append(g)
            while len(issued) <= g:
                issue(len(issued))

        error = None
        try:
            model.backward(on_group=on_group)
        except BaseException as e:                   # noqa: B902 -- re-raised below, after the collectives are matched
            error = e
        try:
            while len(issued) < len(ranges):         # groups the sweep never announced (only after an error)
                if error is None:
                    out_of_turn.append(len(issued))
                issue(len(issued))
        finally:
            if defer:
                model.set_group_join(True)
            if cuda:
                with torch.cuda.stream(self._stream):
                    for g, w in works:
                        w.wait()
                        a, b = ranges[g]
                        summed = self._rsag[g]['full'][:b - a] if rs_ag else (send[a:b] if bf16 else None)
                        if bf16:                      # the summed bf16 slice back into the fp32 gradient buffer (the optimizer's input)
                            _unpack_bf16(summed, flat[a:b], self._stream)
                        elif summed is not None:
                            flat[a:b].copy_(summed)
                main.wait_stream(self._stream)
            else:
                for g, w in works:
                    w.wait()
                    a, b = ranges[g]
                    if rs_ag:
                        flat[a:b].copy_(self._rsag[g]['full'][:b - a])
                    elif bf16:
                        flat[a:b].copy_(send[a:b])
        if error is not None:
            raise error
        if out_of_turn:
            raise RuntimeError('gradient groups were announced out of order or not at all: %r (collectives were still issued '
                               'in ascending order; this step\'s gradients are not valid)' % (out_of_turn,))
        return flat


class _NullContext(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _pack_bf16(src, dst, stream):
    """fp32 device slice -> bf16 device slice on `stream` (csrc/backward.hip: grad_pack_bf16_kernel)."""
    from . import _lib
    lib = _lib.load()
    _lib.check(lib.pivp_grad_pack_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), stream.cuda_stream), 'pivp_grad_pack_bf16')


def _sum_shards(recv, nshards, shard_len, out, stream):
    """out[i] = sum_k recv[k * shard_len + i] accumulated in fp32 in the order k = 0 .. nshards-1, rounded once to out's type
    (csrc/backward.hip: grad_sum_shards_kernel)."""
    from . import _lib
    lib = _lib.load()
    _lib.check(lib.pivp_grad_sum_shards(recv.data_ptr(), 1 if recv.dtype == torch.bfloat16 else 0, int(nshards), int(shard_len), out.data_ptr(),
                                        1 if out.dtype == torch.bfloat16 else 0, stream.cuda_stream), 'pivp_grad_sum_shards')


def _unpack_bf16(src, dst, stream):
    from . import _lib
    lib = _lib.load()
    _lib.check(lib.pivp_grad_unpack_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), stream.cuda_stream), 'pivp_grad_unpack_bf16')


def shard_batch(arrays, rank, world_size):
    """Contiguous per-rank shard of time-major arrays (T, B, ...) along the batch axis."""
    out = []
    for a in arrays:
        B = a.shape[1]
        if B % world_size:
            raise ValueError('global batch %d is not divisible by %d ranks' % (B, world_size))
        per = B // world_size
        out.append(a[:, rank * per:(rank + 1) * per])
    return out
