"""Per-tap error of the HIP path vs the fp64 oracle (and the fp32 oracle's own error) at chosen steps."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import pivp_amd
from oracle import restatement as R

mt = sys.argv[1] if len(sys.argv) > 1 else 'CDNA'
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nm = 1 if mt == 'DNA' else 10
kw = dict(is_cdna=mt == 'CDNA', is_stp=mt == 'STP', is_dna=mt == 'DNA')
P = R.init_params(seed=1, dtype=np.float64, scale=1.0, num_masks=nm, model_type=mt)
imgs, acts, stas = R.synthetic_batch(2, T)
steps = tuple(range(T - 1))
ref = R.Model(nm, params=P, dtype=np.float64, prefix='x', **kw); ref.train = False
ref([imgs, acts, stas], 0, tap_steps=steps)
r32 = R.Model(nm, params=P, dtype=np.float32, prefix='x', **kw); r32.train = False
r32([imgs, acts, stas], 0, tap_steps=steps)
m = pivp_amd.Model(nm, prefix='x', keep_activations=True, **kw)
m.load_state_dict_reference(P)
with pivp_amd.using_config('train', False):
    m([imgs, acts, stas], 0)
gen = torch.stack(m.gen_images).cpu().numpy()
names = ['enc0', 'hidden1', 'hidden2', 'enc1', 'hidden3', 'hidden4', 'enc2', 'enc3', 'hidden5', 'enc4', 'hidden6', 'enc5', 'hidden7', 'enc6']
for t in steps:
    print('step', t)
    for n in names:
        a = m.tap(n, t).cpu().numpy(); b = ref.taps[t][n]; c = r32.taps[t][n]
        print('  %-8s hip max %.2e rms %.2e | np32 max %.2e rms %.2e | scale %.2f' % (
            n, np.abs(a - b).max(), np.sqrt(((a - b) ** 2).mean()), np.abs(c - b).max(), np.sqrt(((c - b) ** 2).mean()), np.abs(b).mean()))
    l2 = R.per_pixel_l2(gen[t], ref.gen_images[t]); l32 = R.per_pixel_l2(r32.gen_images[t], ref.gen_images[t])
    print('  output   hip max %.2e rms %.2e | np32 max %.2e rms %.2e' % (l2.max(), np.sqrt((l2**2).mean()), l32.max(), np.sqrt((l32**2).mean())))
