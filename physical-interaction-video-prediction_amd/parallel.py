"""Data parallelism over the batch axis: one process per GPU, replicated parameters, one sum all-reduce of the flat
gradient buffer per step (the reference itself is single-device: its `optimizer.update`, TM:950, is the
call this wraps).

On MI355X the backend is "nccl" (= RCCL over xGMI); the same code runs with "gloo" on CPU tensors, which is how the
host logic -- including the overlapped, per-group path -- is tested without a GPU (tests/test_parallel_gloo.py)."""
import torch
import torch.distributed as dist


class GradAllReduce(object):
    """Sum-all-reduce of a flat gradient buffer in `nbuckets` contiguous slices on a side stream.

    xGMI is a full mesh of point-to-point links, so a few large slices (not many small ones) keep every link busy;
    the default of 4 buckets of ~9 MB each lets the first slices travel while the tail of backward still runs when the
    caller invokes `allreduce_range` per finished region (the plan fills gradients back-to-front)."""

    def __init__(self, group=None, nbuckets=4, defer_side_wait=True, payload='auto'):
        """payload: what travels in the overlapped all-reduce (`backward_and_allreduce`).  'fp32' = the flat gradient buffer itself
        (36.9 MB per step at 64x64).  'bf16' = every announced group slice is rounded to bf16 into a send buffer on the collective's
        stream (one HIP launch), the bf16 buffer is all-reduced (18.4 MB) and the sum is written back into the fp32 flat buffer,
        which stays the optimizer's input ("bf16 payload + fp32 master accumulation").  'auto' (default) = 'bf16'
        for a model in the bf16 precision mode (BASELINE.json config 3), 'fp32' otherwise; pass 'fp32' to opt out."""
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        if payload not in ('auto', 'fp32', 'bf16'):
            raise ValueError("payload must be 'auto', 'fp32' or 'bf16'")
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
        bf16 = self.payload == 'bf16' or (self.payload == 'auto' and getattr(model, 'precision', 'fp32') == 'bf16')
        send = None
        if bf16:
            if self._send is None or self._send.numel() != flat.numel() or self._send.device != flat.device:
                self._send = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
            send = self._send
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
                    if bf16:
                        _pack_bf16(flat[a:b], send[a:b], self._stream)
                    works.append((g, dist.all_reduce(send[a:b] if bf16 else flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
            else:
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
                        if bf16:                      # the summed bf16 slice back into the fp32 gradient buffer (the optimizer's input)
                            a, b = ranges[g]
                            _unpack_bf16(send[a:b], flat[a:b], self._stream)
                main.wait_stream(self._stream)
            else:
                for g, w in works:
                    w.wait()
                    if bf16:
                        a, b = ranges[g]
                        flat[a:b].copy_(send[a:b])
        if error is not None:
            raise error
        if out_of_turn:
            raise RuntimeError('gradient groups were announced out of order or not at all: %r (collectives were still issued '
                               'in ascending order; this step\'s gradients are not valid)' % (out_of_turn,))
        return flat


def _pack_bf16(src, dst, stream):
    """fp32 device slice -> bf16 device slice on `stream` (csrc/backward.hip: grad_pack_bf16_kernel)."""
    from . import _lib
    lib = _lib.load()
    _lib.check(lib.pivp_grad_pack_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), stream.cuda_stream), 'pivp_grad_pack_bf16')


def _unpack_bf16(src, dst, stream):
    from . import _lib
    lib = _lib.load()
    _lib.check(lib.pivp_grad_unpack_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), stream.cuda_stream), 'pivp_grad_unpack_bf16')


class HostStubModel(object):
    """CPU stand-in for `Model` with the same training protocol (`_ensure_grads`, `grad_group_ranges`, `cleargrads`,
    `backward(on_group)`): the gradient of group g is `value * (g + 1)` everywhere.  It lets the data-parallel host logic run
    under gloo with no GPU: tests/test_parallel_gloo.py and `bench.py --dry`.  `fail_in_group` makes the callback of that group
    raise, as a failing rank would; `skip_groups` leaves groups unannounced."""

    def __init__(self, sizes=(1000, 300, 70, 5000, 64, 1), value=1.0, fail_in_group=None, skip_groups=(), precision='fp32'):
        self.precision = precision
        self.sizes = list(sizes)
        self.value = float(value)
        self.fail_in_group = fail_in_group
        self.skip_groups = set(skip_groups)
        self._flat_params = torch.zeros(sum(self.sizes))
        self._flat = torch.zeros(sum(self.sizes))
        self.announced = []

    def _ensure_grads(self):
        return self._flat

    def cleargrads(self):
        self._flat.zero_()

    def grad_group_ranges(self):
        out, o = [], 0
        for n in self.sizes:
            out.append((o, o + n))
            o += n
        return out

    def backward(self, on_group=None):
        errors = []
        for g, (a, b) in enumerate(self.grad_group_ranges()):
            self._flat[a:b] += self.value * (g + 1)
            if on_group is None or g in self.skip_groups:
                continue
            self.announced.append(g)
            try:                                     # like Model.backward: the sweep continues, the first error is raised afterwards
                if g == self.fail_in_group:
                    raise RuntimeError('injected failure in group %d' % g)
                on_group(g)
            except BaseException as e:               # noqa: B902
                errors.append(e)
        if errors:
            raise errors[0]


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
