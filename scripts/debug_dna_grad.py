"""Where does the DNA gradient check lose accuracy?  Prints, per tensor, the error distribution against autograd."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import pivp_amd
from oracle import restatement as R
from oracle.torch_restatement import TorchModel

P = R.init_params(seed=1, dtype=np.float64, scale=1.0, model_type='DNA', num_masks=1)
imgs, acts, stas = R.synthetic_batch(2, 4)
tm = TorchModel(1, params=P, requires_grad=True, scheduled_sampling_k=-1, is_cdna=False, is_dna=True)
loss = tm([imgs, acts, stas], 0); loss.backward()
gref = {k: v.grad.numpy() for k, v in tm.p.items()}
m = pivp_amd.Model(1, is_cdna=False, is_dna=True, prefix='t', keep_activations=True)
m.load_state_dict_reference(P)
m([imgs, acts, stas], 0); m.cleargrads(); m.backward()
got = m.grads_reference()
for k in ('norm_enc6/norm/beta', 'norm_enc6/norm/gamma', 'enc6/b', 'masks/W'):
    g = gref[k]; e = np.abs(got[k].astype(np.float64) - g); sc = np.abs(g).max()
    order = np.argsort(e.ravel())[::-1][:6]
    print(k, 'scale %.3e' % sc, 'n', g.size, 'over 1e-3*scale:', int((e > 1e-3 * sc).sum()))
    for i in order:
        print('   idx %d  err %.3e (rel %.3e)  ref %.4e got %.4e' % (i, e.ravel()[i], e.ravel()[i] / sc, g.ravel()[i], got[k].ravel()[i]))
