"""Experiment: would the rollout gain from running the batch as two halves on two streams (the small latency-bound kernels of one half
under the ConvLSTM kernels of the other)?  Two independent B/2 models on two streams, enqueued from two host threads, against one B model."""
import sys, time, threading
import numpy as np, torch
sys.path.insert(0, '.')
import pivp_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PREC = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
steps = 20
rs = np.random.RandomState(0)
def batch(b):
    return [torch.from_numpy(rs.random_sample((10, b, 3, 64, 64)).astype(np.float32)).cuda(),
            torch.from_numpy((0.1 * rs.standard_normal((10, b, 5))).astype(np.float32)).cuda(),
            torch.from_numpy((0.1 * rs.standard_normal((10, b, 5))).astype(np.float32)).cuda()]
np.random.seed(0)
def timed(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with pivp_amd.using_config('train', False):
    full = pivp_amd.Model(10, prefix='x', precision=PREC); xf = batch(B)
    def run_full(): full.reset_state(); full(xf, 0)
    t_full = timed(run_full, steps)
    halves = [pivp_amd.Model(10, prefix='x', precision=PREC) for _ in range(2)]; xh = [batch(B // 2) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    def run_half(i):
        with torch.cuda.stream(streams[i]):
            halves[i].reset_state(); halves[i](xh[i], 0)
    def run_seq():
        run_half(0); run_half(1)
    t_seq = timed(run_seq, steps)          # one host thread enqueues both streams
    def run_thr():
        th = [threading.Thread(target=run_half, args=(i,)) for i in range(2)]
        for t in th: t.start()
        for t in th: t.join()
    t_thr = timed(run_thr, steps)
    one = pivp_amd.Model(10, prefix='x', precision=PREC)
    def run_one(): one.reset_state(); one(xh[0], 0)
    t_one = timed(run_one, steps)
print('B=%d: one model %.3f ms | one B/2 model alone %.3f ms | two B/2 models on two streams: enqueued by one thread %.3f ms, by two threads %.3f ms' % (B, t_full, t_one, t_seq, t_thr))
