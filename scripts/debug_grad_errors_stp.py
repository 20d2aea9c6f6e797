"""Per-tensor gradient errors of the HIP backward vs float64 autograd for the STP test case (tests/test_gpu_train.py::test_bptt_gradients_stp)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import pivp_amd
from oracle import restatement as R
from test_gpu_train import _autograd
P = R.init_params(seed=1, dtype=np.float64, scale=1.0, model_type='STP')
imgs, acts, stas = R.smooth_batch(2, 4)
loss_ref, gref = _autograd(P, imgs, acts, stas, is_cdna=False, is_stp=True)
m = pivp_amd.Model(10, is_cdna=False, is_stp=True, prefix='t', keep_activations=True)
m.load_state_dict_reference(P)
loss = float(m([imgs, acts, stas], 0))
m.cleargrads(); m.backward()
got = m.grads_reference()
rows = []
for k, g in gref.items():
    scale = np.abs(g).max() + 1e-12
    e = np.abs(got[k].astype(np.float64) - g).ravel() / scale
    es = np.sort(e)
    l2 = np.linalg.norm(got[k].astype(np.float64) - g) / (np.linalg.norm(g) + 1e-30)
    rows.append((es[-1], es[-9] if e.size > 9 else 0, np.median(e), scale, k, e.size, l2, es[int(0.999 * (e.size - 1))]))
rows.sort(reverse=True)
for r in rows[:16]:
    print('%-28s max %.2e 9th %.2e p99.9 %.2e median %.2e scale %.2e n %d relL2 %.2e' % (r[4], r[0], r[1], r[7], r[2], r[3], r[5], r[6]))
k = 'hidden6/norm/gamma'
g = gref[k]; e = np.abs(got[k].astype(np.float64) - g).ravel() / (np.abs(g).max() + 1e-12)
idx = np.argsort(e)[-16:][::-1]
print(k, [(int(i // 256), int(i % 256 // 16), int(i % 16), float('%.2e' % e[i])) for i in idx])
