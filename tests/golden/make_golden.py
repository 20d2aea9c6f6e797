"""Generate the committed golden fixtures from the float64 NumPy oracle.

Run from the repo root:  python tests/golden/make_golden.py
The reference itself cannot run here (Chainer 2.0.1 / Python 2 absent), so these
vectors pin HIP <-> oracle; oracle <-> reference is pinned by the KATs and the line map.
Weights are NOT stored (36.85 MB): they are regenerated from `init_params(seed=1, scale=1.0)`,
which uses numpy's frozen legacy RandomState stream; a checksum of them is stored instead.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import restatement as R  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
TAP_STRIDE = 97


def run(model_type, num_masks, batch, seq_len, tap_steps):
    P = R.init_params(seed=1, dtype=np.float64, scale=1.0, num_masks=num_masks, model_type=model_type)
    imgs, acts, stas = R.synthetic_batch(batch, seq_len, seed=0)
    m = R.Model(num_masks, is_cdna=model_type == 'CDNA', is_stp=model_type == 'STP',
                is_dna=model_type == 'DNA', params=P, dtype=np.float64, prefix='golden')
    m.train = False
    loss = m([imgs, acts, stas], 0, tap_steps=tap_steps)
    out = dict(loss=np.float64(loss), psnr_all=np.float64(m.psnr_all),
               gen_images=np.stack(m.gen_images).astype(np.float32),
               gen_states=np.stack(m.gen_states).astype(np.float32),
               param_checksum=np.float64(sum(float(np.abs(v).sum()) for v in P.values())),
               batch=batch, seq_len=seq_len, num_masks=num_masks)
    for t, taps in m.taps.items():
        for name in ('enc0', 'enc1', 'enc2', 'enc3', 'enc4', 'enc5', 'enc6', 'enc7', 'hidden5', 'masks'):
            out['tap%d_%s' % (t, name)] = taps[name].ravel()[::TAP_STRIDE].astype(np.float32)
    if model_type == 'CDNA':
        out['cdna_kerns_last'] = m.last_cdna_kerns.astype(np.float32)
    return out


if __name__ == '__main__':
    np.savez_compressed(os.path.join(OUT, 'cdna_b2_t10.npz'), **run('CDNA', 10, 2, 10, (0, 8)))
    np.savez_compressed(os.path.join(OUT, 'stp_b2_t4.npz'), **run('STP', 10, 2, 4, (0, 2)))
    np.savez_compressed(os.path.join(OUT, 'dna_b2_t4.npz'), **run('DNA', 1, 2, 4, (0, 2)))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
