"""Per-timestep error of the bf16 precision mode against the float64 oracle's golden rollout (config 1: B = 2, T = 10, CDNA,
white-noise frames, random-init weights): what `--precision bf16` "reports instead of meeting the 1e-4 gate" (SURVEY.md 8d)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import pivp_amd
from oracle import restatement as R
g = np.load(os.path.join('tests', 'golden', 'cdna_b2_t10.npz'))
P = R.init_params(seed=1, dtype=np.float32, scale=1.0, num_masks=10, model_type='CDNA')
imgs, acts, stas = R.synthetic_batch(2, 10)
for prec in ('fp32', 'bf16x3', 'bf16'):
    m = pivp_amd.Model(10, prefix='r', precision=prec)
    m.load_state_dict_reference(P)
    with pivp_amd.using_config('train', False):
        loss = float(m([imgs, acts, stas], 0))
    gen = torch.stack(m.gen_images).cpu().numpy()
    l2 = R.per_pixel_l2(gen, g['gen_images'])            # [T-1][B][H][W]
    per_t = ['%.1e/%.1e' % (l2[t].max(), np.sqrt((l2[t] ** 2).mean())) for t in range(l2.shape[0])]
    print('%s: loss %.6f (oracle %.6f); per-pixel L2 max/rms by predicted frame: %s' % (prec, loss, float(g['loss']), ' '.join(per_t)))

# the other heads on their golden fixtures (T = 4): max / rms over all frames
for name, mt, nm in (('stp_b2_t4', 'STP', 10), ('dna_b2_t4', 'DNA', 1)):
    g = np.load(os.path.join('tests', 'golden', name + '.npz'))
    P = R.init_params(seed=1, dtype=np.float32, scale=1.0, num_masks=nm, model_type=mt)
    imgs, acts, stas = R.synthetic_batch(int(g['batch']), int(g['seq_len']))
    for prec in ('fp32', 'bf16x3', 'bf16'):
        m = pivp_amd.Model(nm, is_cdna=False, is_stp=mt == 'STP', is_dna=mt == 'DNA', prefix='r', precision=prec)
        m.load_state_dict_reference(P)
        with pivp_amd.using_config('train', False):
            m([imgs, acts, stas], 0)
        l2 = R.per_pixel_l2(torch.stack(m.gen_images).cpu().numpy(), g['gen_images'])
        print('%s %s: per-pixel L2 max %.1e rms %.1e' % (mt, prec, l2.max(), np.sqrt((l2 ** 2).mean())))
