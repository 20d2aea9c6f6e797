"""Does the fp32 ConvLSTM weight-gradient kernel's duration depend on the VALUES of its operands?  (igemm_wgrad.hip's ablations: staging loads
and LDS stores cost nothing on their own and 43 us together, i.e. what costs is LDS contents that change.)  lstm7's cell backward at B = 32
with random / zero activations (x, h) and random / zero incoming gradients (-> dG); run under rocprofv3 --kernel-trace --stats, one
variant per process:  python scripts/wgrad_data_dependence.py --x random|zero --dg random|zero"""
import argparse, sys
import numpy as np, torch
sys.path.insert(0, '.')
import pivp_amd
from pivp_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument('--x', default='random'); ap.add_argument('--dg', default='random'); ap.add_argument('--reps', type=int, default=30)
args = ap.parse_args()
lib = _lib.load(); dev = 'cuda:0'
B, cx, C, H = 32, 96, 32, 32
M, cin, N = B * H * H, cx + C, 4 * C
st = torch.cuda.current_stream().cuda_stream
mk = (lambda *s: torch.randn(*s, device=dev)) if args.x == 'random' else (lambda *s: torch.zeros(*s, device=dev))
x, h = mk(M, cx), mk(M, C) * 0.5
w = torch.randn(25 * cin * N, device=dev) / np.sqrt(25 * cin)
gates = torch.rand(M, N, device=dev) * 0.8 + 0.1; c_old = torch.randn(M, C, device=dev); c_new = torch.randn(M, C, device=dev)
dmk = (lambda *s: torch.randn(*s, device=dev)) if args.dg == 'random' else (lambda *s: torch.zeros(*s, device=dev))
dh, dc0 = dmk(M, C), dmk(M, C)
dG = torch.empty(M, N, device=dev); wt = torch.empty_like(w); d_in = torch.empty(M, cin, device=dev)
dW = torch.zeros_like(w); db = torch.zeros(N, device=dev)
for _ in range(args.reps):
    dc = dc0.clone()
    _lib.check(lib.pivp_convlstm_backward(x.data_ptr(), cx, cx, h.data_ptr(), C, w.data_ptr(), gates.data_ptr(), c_old.data_ptr(), c_new.data_ptr(),
                                          dh.data_ptr(), C, None, 0, dc.data_ptr(), 1, dG.data_ptr(), wt.data_ptr(), d_in.data_ptr(), dW.data_ptr(), db.data_ptr(),
                                          B, H, H, st), 'bwd')
torch.cuda.synchronize()
print('x %s, dG %s: |dG| max %.3g' % (args.x, args.dg, float(dG.abs().max())))
