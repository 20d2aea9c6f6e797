"""Data parallelism over the batch axis: one process per GPU, replicated parameters, one sum all-reduce of the flat
gradient buffer per step (the reference itself is single-device).

On MI355X the backend is "nccl" (= RCCL over xGMI); the same code runs with "gloo" on CPU tensors, which is how the
host logic is tested without a GPU (tests/test_parallel_gloo.py)."""
import torch
import torch.distributed as dist


class GradAllReduce(object):
    """Sum-all-reduce of a flat gradient buffer in `nbuckets` contiguous slices on a side stream.

    xGMI is a full mesh of point-to-point links, so a few large slices (not many small ones) keep every link busy;
    the default of 4 buckets of ~9 MB each lets the first slices travel while the tail of backward still runs when the
    caller invokes `allreduce_range` per finished region (the plan fills gradients back-to-front)."""

    def __init__(self, group=None, nbuckets=4):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.nbuckets = max(1, int(nbuckets))
        self._stream = None

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
        """model.backward() with the all-reduce of each gradient group launched on the side stream as soon as the sweep
        has finished that group at t = 0 (every parameter is shared by all timesteps, so nothing is final earlier): the
        heads / enc6 slice travels while lstm7 .. enc0 of the last timestep are still being differentiated.  Six
        contiguous slices of 1-11 MB (model.grad_group_ranges()); xGMI is point-to-point, a few large messages keep
        its links busier than many small ones."""
        flat = model._ensure_grads()
        if (self.world_size == 1 and not force_overlap) or not flat.is_cuda:   # force_overlap: exercise the path on one rank (tests)
            model.backward()
            return self.allreduce_flat(flat)
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=flat.device)
        ranges = model.grad_group_ranges()
        main = torch.cuda.current_stream(flat.device)
        works = []

        def on_group(g):
            a, b = ranges[g]
            ev = torch.cuda.Event()
            ev.record(main)                      # everything that writes slice g is already enqueued on `main`
            self._stream.wait_event(ev)
            with torch.cuda.stream(self._stream):
                works.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

        model.backward(on_group=on_group)
        with torch.cuda.stream(self._stream):
            for w in works:
                w.wait()
        main.wait_stream(self._stream)
        return flat


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
