import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import pivp_amd
from oracle import restatement as R
P = R.init_params(seed=1, dtype=np.float32, scale=1.0)
imgs, acts, stas = R.synthetic_batch(32, 10)
for side in ('0', '1'):
    os.environ['PIVP_SIDE_STREAM'] = side
    m = pivp_amd.Model(10, prefix='s', keep_activations=True, precision='bf16')
    m.load_state_dict_reference(P)
    out = []; gens = []
    for _ in range(5):
        m.reset_state()
        with pivp_amd.using_config('train', True):
            m([imgs, acts, stas], 0)
            gens.append(torch.stack(m.gen_images).clone())
            m.cleargrads(); m.backward()
        out.append(m._flat_grads.clone())
    torch.cuda.synchronize()
    print('side', side, 'forward identical across sweeps:', [bool(torch.equal(g, gens[0])) for g in gens],
          'grad rel diff vs sweep 0:', ['%.1e' % float((g - out[0]).norm() / out[0].norm()) for g in out])
    if side == '0': ref = out
    else: print('side 1 vs side 0 per sweep:', ['%.1e' % float((a - b).norm() / b.norm()) for a, b in zip(out, ref)])
