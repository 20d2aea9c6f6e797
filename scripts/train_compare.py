"""The same 150 Adam steps (config 2's shapes at B = 8: CDNA, T = 10, 64 x 64, random-init weights, one fixed synthetic batch, feed-self) in the fp32
kernels and in the split modes: do the loss curves of fp32-grade arithmetic stay together?  (A train step's gradient is within 5e-5 of the fp32 kernels';
this asks what 150 optimizer steps make of that.)   python scripts/train_compare.py [steps]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import pivp_amd
from oracle import restatement as R

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B, T = 8, 10
P = R.init_params(seed=3, dtype=np.float32, scale=1.0)
imgs, acts, stas = R.synthetic_batch(B, T)
# video-like frames (box blur of the noise) so that there is something to learn
from numpy.lib.stride_tricks import sliding_window_view
pad = np.pad(imgs, ((0, 0), (0, 0), (0, 0), (5, 5), (5, 5)), mode='reflect')
imgs = np.ascontiguousarray(sliding_window_view(pad, (11, 11), axis=(3, 4)).mean(axis=(-1, -2))).astype(np.float32)
curves = {}
for prec in ('fp32', 'fp16x3', 'bf16x6', 'bf16'):
    m = pivp_amd.Model(10, prefix='t', precision=prec, keep_activations=True)
    m.load_state_dict_reference(P)
    op = pivp_amd.Adam(alpha=0.001).setup(m)
    losses = []
    for it in range(steps):
        m.reset_state()
        losses.append(float(op.update(m, [imgs, acts, stas], 0)))
    curves[prec] = np.array(losses)
    print('%-7s loss: step 0 %.6f, 10 %.6f, 50 %.6f, last %.6f; finite %s' % (prec, losses[0], losses[10], losses[min(50, steps - 1)], losses[-1], np.isfinite(losses).all()))
ref = curves['fp32']
for prec in ('fp16x3', 'bf16x6', 'bf16'):
    rel = np.abs(curves[prec] - ref) / ref
    print('%-7s against the fp32 kernels: relative loss difference max %.2e (step %d), at the last step %.2e, mean over the last 20 steps %.2e'
          % (prec, rel.max(), int(rel.argmax()), rel[-1], rel[-20:].mean()))
