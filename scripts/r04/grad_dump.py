"""Gradient of one train-mode rollout in the fp16x3 mode (B = 2, T = 10, config 1's inputs) written to a file: scripts/r04/call69.sh runs it under
PIVP_X3_DGRAD / PIVP_X3_WGRAD = 0 / 1 (read once per process) and compares the files -- do the switches reach the kernels they name?"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import pivp_amd
from oracle import restatement as R

out = sys.argv[1]
prec = sys.argv[2] if len(sys.argv) > 2 else 'fp16x3'
P = R.init_params(seed=1, dtype=np.float32, scale=1.0)
imgs, acts, stas = R.synthetic_batch(2, 10)
m = pivp_amd.Model(10, prefix='t', precision=prec, keep_activations=True)
m.load_state_dict_reference(P)
with pivp_amd.using_config('train', True):
    loss = m([imgs, acts, stas], 0)
m.cleargrads()
m.backward()
torch.cuda.synchronize()
g = m._flat_grads.detach().cpu().numpy()
np.save(out, g)
print(out, float(loss), float(np.linalg.norm(g)))
