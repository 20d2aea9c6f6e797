"""All parameter gradients of a small rollout against float64 autograd, twice: relative L2 error per tensor and the run-to-run spread (what
fp32 atomics order explains is ~1e-6; anything larger is a race or a stale read).  python scripts/grad_spread_report.py [CDNA|STP|DNA] [num_masks]"""
import sys

import numpy as np

sys.path.insert(0, '.')
from oracle import restatement as R
from oracle.torch_restatement import TorchModel
import pivp_amd as pivp

mt = sys.argv[1] if len(sys.argv) > 1 else 'CDNA'
nm = int(sys.argv[2]) if len(sys.argv) > 2 else (1 if mt == 'DNA' else 10)
kinds = dict(is_cdna=mt == 'CDNA', is_stp=mt == 'STP', is_dna=mt == 'DNA')
P = R.init_params(seed=2, dtype=np.float64, scale=1.0, num_masks=nm, model_type=mt)
imgs, acts, stas = R.synthetic_batch(3, 5)
tm = TorchModel(nm, params=P, requires_grad=True, **kinds)
tm([imgs, acts, stas], 0).backward()
runs = []
for r in range(2):
    m = pivp.Model(nm, prefix='t', keep_activations=True, **kinds)
    m.load_state_dict_reference(P)
    m([imgs, acts, stas], 0); m.cleargrads(); m.backward()
    runs.append({k: v.astype(np.float64) for k, v in m.grads_reference().items()})
rows = []
for k, v in tm.p.items():
    g = v.grad.numpy()
    n = np.linalg.norm(g) + 1e-30
    rows.append((np.linalg.norm(runs[0][k] - runs[1][k]) / n, np.linalg.norm(runs[0][k] - g) / n, k))
rows.sort(reverse=True)
print('%s num_masks=%d: worst run-to-run spread %.2e (%s); worst error %.2e (%s)' % (mt, nm, rows[0][0], rows[0][2], max(r[1] for r in rows),
                                                                                   max(rows, key=lambda r: r[1])[2]))
for sp, er, k in rows[:6]:
    print('   %-28s spread %.2e  error %.2e' % (k, sp, er))
