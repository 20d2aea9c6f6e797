"""HIP path against the random-init float64 fixtures: max / rms per-pixel L2 for each fixture directory given.

    python scripts/fixture_distances.py tests/golden [other_dir ...]

Round 5 regenerated the random-init fixtures with the float64 oracle fed the fp32-rounded parameters (oracle/restatement.py:
init_params_widened) instead of the unrounded float64 draw; run with the old fixtures in a second directory this prints the
before / after distances of the SAME HIP outputs (profiles/r05/fixture_distances.txt)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pivp_amd  # noqa: E402
from oracle import restatement as R  # noqa: E402

SMALL = [('cdna_b2_t10', 'CDNA', 10), ('stp_b2_t4', 'STP', 10), ('dna_b2_t4', 'DNA', 1)]
FULL = [('cdna_b32_t10', 'CDNA', 64, False), ('stp_b32_t10', 'STP', 64, False), ('stp_b32_t10_smooth', 'STP', 64, True),
        ('cdna_128_b2_t20', 'CDNA', 128, False)]


def rollout(mt, nm, B, T, size, smooth, precision):
    P = R.init_params(seed=1, dtype=np.float32, scale=1.0, num_masks=nm, model_type=mt, height=size, width=size)
    imgs, acts, stas = (R.smooth_batch if smooth else R.synthetic_batch)(B, T, size, size, seed=0)
    m = pivp_amd.Model(nm, is_cdna=mt == 'CDNA', is_stp=mt == 'STP', is_dna=mt == 'DNA', prefix='d', precision=precision)
    m.load_state_dict_reference(P)
    with pivp_amd.using_config('train', False):
        loss = float(m([imgs, acts, stas], 0))
    return loss, torch.stack(m.gen_images).cpu().numpy()


def main(dirs):
    outs = {}
    for precision in ('fp32', 'bf16x6', 'fp16x3'):
        for name, mt, nm in SMALL:
            g0 = np.load(os.path.join(dirs[0], name + '.npz'))
            outs[(name, precision)] = rollout(mt, nm, int(g0['batch']), int(g0['seq_len']), 64, False, precision)
        for name, mt, size, smooth in FULL:
            g0 = np.load(os.path.join(dirs[0], name + '.npz'))
            outs[(name, precision)] = rollout(mt, 10, int(g0['batch']), int(g0['seq_len']), size, smooth, precision)
    print('%-22s %-7s %-28s %12s %12s %12s %14s' % ('fixture', 'mode', 'directory', 'max L2', 'max L2 t<2', 'rms L2', '|loss diff|'))
    for (name, precision), (loss, gen) in outs.items():
        for d in dirs:
            g = np.load(os.path.join(d, name + '.npz'))
            if 'gen_images' in g.files:
                l2 = R.per_pixel_l2(gen, g['gen_images'])                      # (T-1, B, H, W)
                first = l2[:2].max()
            else:                                                              # full-size fixtures keep every pixel_stride-th pixel
                pix = np.ascontiguousarray(gen.transpose(0, 1, 3, 4, 2)).reshape(-1, 3)[::int(g['pixel_stride'])]
                l2 = np.sqrt(((pix.astype(np.float64) - g['gen_pixels']) ** 2).sum(axis=1))
                per_step = gen.shape[1] * gen.shape[3] * gen.shape[4]          # pixels per step before the stride
                n2 = (2 * per_step + int(g['pixel_stride']) - 1) // int(g['pixel_stride'])
                first = l2[:n2].max()
            print('%-22s %-7s %-28s %12.3e %12.3e %12.3e %14.3e' % (name, precision, d, l2.max(), first, np.sqrt((l2 ** 2).mean()),
                                                                    abs(loss - float(g['loss']))))


if __name__ == '__main__':
    main(sys.argv[1:] or [os.path.join(ROOT, 'tests', 'golden')])
