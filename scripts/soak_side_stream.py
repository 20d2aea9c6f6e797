"""Soak check of the backward sweep's side stream (weight gradients beside the critical path): at config 2's size the flat gradient of
N repeated sweeps must agree with a sweep that keeps everything on one stream to fp32-atomics noise.  A missing join (a dY buffer
rewritten while a weight gradient still reads it) would show as an outlier."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
prec = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
import pivp_amd
from oracle import restatement as R
P = R.init_params(seed=1, dtype=np.float32, scale=1.0)
imgs, acts, stas = R.synthetic_batch(32, 10)
def grads(side):
    os.environ['PIVP_SIDE_STREAM'] = side
    m = pivp_amd.Model(10, prefix='s', keep_activations=True, precision=prec)
    m.load_state_dict_reference(P)
    out = []
    for _ in range(N if side == '1' else 1):
        m.reset_state()
        with pivp_amd.using_config('train', True):
            m([imgs, acts, stas], 0)
            m.cleargrads(); m.backward()
        out.append(m._flat_grads.clone())
    torch.cuda.synchronize()
    return out
# bf16 mode: the order of the K-split data gradients' atomics moves dG by ~1e-7, and rounding dG to bf16 in the next kernel turns that into
# ~1.5e-4 of the flat gradient from sweep to sweep, with or without the side stream (scripts/soak_bf16_sweeps.py); fp32: 1e-7
tol = 1e-5 if prec == 'fp32' else 1e-3
ref = grads('0')[0]
worst = 0.0
for i, g in enumerate(grads('1')):
    rel = float((g - ref).norm() / ref.norm()); mx = float((g - ref).abs().max() / ref.abs().max())
    worst = max(worst, rel)
    if rel > tol or not bool(torch.isfinite(g).all()):
        print('OUTLIER at sweep', i, rel, mx)
print('%s: %d sweeps with the side stream vs one without: worst relative L2 difference %.2e' % (prec, N, worst))
assert worst < tol
