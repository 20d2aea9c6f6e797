"""Weight gradients of the five stride-2 3x3 layers at the B = 32 shapes of config 2: T - 1 = 9 timesteps as nine launches into the partial
planes + one reduction (how the sweep ran them through round 4), and as ONE batched launch + the reduction (hipEvent timing through the
C ABI, random operands; both forms are checked against each other).  Usage: python scripts/bench_enc_wgrad.py [B] [iters] [T]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import pivp_amd
from pivp_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T = int(sys.argv[3]) if len(sys.argv) > 3 else 9
lib = _lib.load()
dev = 'cuda:0'
st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev); gen.manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=gen)

def timed(f):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

print('| layer | 9 launches + reduction, us | one batched launch + reduction, us | reduction alone, us | GFLOP | TFLOP/s batched | max rel diff |')
print('|---|---|---|---|---|---|---|')
for name, mode, cin, cout, H in (('enc6', 1, 64, 64, 32), ('enc5', 1, 96, 96, 16), ('enc4', 1, 128, 128, 8), ('enc2', 0, 64, 64, 16), ('enc1', 0, 32, 32, 32)):
    Ho = 2 * H if mode else H // 2
    x = R(T, B, H, H, cin); dy = R(T, B, Ho, Ho, cout)
    n = lib.pivp_conv_backward_part_floats(mode, cin, cout, B, H, H)
    part = torch.zeros(n, device=dev); dW1 = torch.zeros(9 * cin * cout, device=dev); dW2 = torch.zeros_like(dW1)
    db1 = torch.zeros(cout, device=dev); db2 = torch.zeros_like(db1)
    xs, ys = x[0].numel() * 4, dy[0].numel() * 4
    def single():
        for t in range(T):
            _lib.check(lib.pivp_conv_wgrad_partial_batch(mode, x.data_ptr() + t * xs, cin, cin, 0, dy.data_ptr() + t * ys, cout, cout, 0, 1, int(t == 0),
                                                         part.data_ptr(), dW1.data_ptr(), db1.data_ptr(), B, H, H, st), 'single')
        _lib.check(lib.pivp_conv_wgrad_partial_reduce(mode, cin, cout, part.data_ptr(), dW1.data_ptr(), db1.data_ptr(), B, H, H, st), 'reduce')
    def batched():
        _lib.check(lib.pivp_conv_wgrad_partial_batch(mode, x.data_ptr() + (T - 1) * xs, cin, cin, -xs, dy.data_ptr(), cout, cout, ys, T, 1,
                                                     part.data_ptr(), dW2.data_ptr(), db2.data_ptr(), B, H, H, st), 'batched')
        _lib.check(lib.pivp_conv_wgrad_partial_reduce(mode, cin, cout, part.data_ptr(), dW2.data_ptr(), db2.data_ptr(), B, H, H, st), 'reduce')
    def red():
        _lib.check(lib.pivp_conv_wgrad_partial_reduce(mode, cin, cout, part.data_ptr(), dW2.data_ptr(), db2.data_ptr(), B, H, H, st), 'reduce')
    single(); torch.cuda.synchronize()
    dW1.zero_(); db1.zero_(); dW2.zero_(); db2.zero_()
    def batched_same():   # same pairing as single(): x[j] with dy[j]
        _lib.check(lib.pivp_conv_wgrad_partial_batch(mode, x.data_ptr(), cin, cin, xs, dy.data_ptr(), cout, cout, ys, T, 1,
                                                     part.data_ptr(), dW2.data_ptr(), db2.data_ptr(), B, H, H, st), 'batched')
        _lib.check(lib.pivp_conv_wgrad_partial_reduce(mode, cin, cout, part.data_ptr(), dW2.data_ptr(), db2.data_ptr(), B, H, H, st), 'reduce')
    single(); batched_same(); torch.cuda.synchronize()
    diff = float((dW1 - dW2).abs().max() / dW1.abs().max()); dbd = float((db1 - db2).abs().max() / db1.abs().max())
    t1 = timed(single); t2 = timed(batched); t3 = timed(red)
    anchors = B * H * H if mode else B * Ho * Ho
    gf = 2.0 * anchors * 9 * cin * cout * T / 1e9
    print('| %s | %.1f | %.1f | %.1f | %.2f | %.1f | %.2e (db %.2e) |' % (name, t1, t2, t3, gf, gf / ((t2 - t3) * 1e-6) / 1e3, diff, dbd))
