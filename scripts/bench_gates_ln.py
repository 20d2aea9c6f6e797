"""The pair "LayerNorm-backward sums + gate backward" of one ConvLSTM cell at config 2's sizes, two-launch form against the form with
the sums inside the gate launch (pivp_gates_backward_ln, fused = 0 / 1).  Run under rocprofv3 --kernel-trace --stats for the
per-kernel durations; by itself it prints the per-call time of the whole entry (which includes the parameter planes' launches)."""
import argparse, sys
import numpy as np, torch
sys.path.insert(0, '.')
import pivp_amd
from pivp_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument('--shape', default='1024,32')          # npix,C of one sample
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--lddy', type=int, default=0)
ap.add_argument('--reps', type=int, default=50)
args = ap.parse_args()
lib = _lib.load()
dev = 'cuda:0'
npix, C = [int(v) for v in args.shape.split(',')]
B, n = args.batch, npix * C
lddy = args.lddy or C
st = torch.cuda.current_stream().cuda_stream
g = torch.rand(B * npix, 4 * C, device=dev); co = torch.randn(B * npix, C, device=dev); cn = torch.randn(B * npix, C, device=dev)
dy = torch.randn(B * npix, lddy, device=dev); gamma = torch.rand(n, device=dev) + 0.5; h = torch.randn(B * npix, C, device=dev) * 0.3
stat = torch.stack((h.reshape(B, -1).mean(1), 1.0 / h.reshape(B, -1).std(1)), 1).contiguous()
dhb = torch.randn(B * npix, C, device=dev); dc = torch.randn(B * npix, C, device=dev); dG = torch.empty(B * npix, 4 * C, device=dev)
dgm = torch.zeros(n, device=dev); dbt = torch.zeros(n, device=dev)
scratch = torch.empty(lib.pivp_gates_backward_ln_scratch_floats(B, n), device=dev)
# something that evicts the operands between calls, as the sweep's other kernels do
junk = torch.empty(64 << 20, device=dev)
for fused in (0, 1):
    if fused and not lib.pivp_gates_backward_ln_fits(B, n, C):
        print('shape not taken by the in-launch sums'); break

    def call():
        junk.add_(1.0)
        _lib.check(lib.pivp_gates_backward_ln(g.data_ptr(), co.data_ptr(), cn.data_ptr(), dy.data_ptr(), lddy, gamma.data_ptr(), stat.data_ptr(),
                                              h.data_ptr(), dhb.data_ptr(), C, dc.data_ptr(), 1, dG.data_ptr(), dgm.data_ptr(), dbt.data_ptr(),
                                              scratch.data_ptr(), B, npix, C, fused, st), 'call')
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        call()
    e1.record(); torch.cuda.synchronize()
    print('npix %d C %d B %d fused %d: %.1f us per call (with the eviction pass and the parameter launches)' % (npix, C, B, fused, e0.elapsed_time(e1) / args.reps * 1e3))
