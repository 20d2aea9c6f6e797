"""Where the STP output error comes from (step t of the stp_b2 fixture): error of hidden5, of theta (HIP's own; float64 math on
HIP's hidden5; float64 math on the oracle's hidden5), and the output error a float64 warp would have with each theta."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import pivp_amd
from oracle import restatement as R

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kw = dict(is_cdna=False, is_stp=True)
P = R.init_params(seed=1, dtype=np.float64, scale=1.0, num_masks=10, model_type='STP')
imgs, acts, stas = R.synthetic_batch(2, T)
steps = tuple(range(T - 1))
ref = R.Model(10, params=P, dtype=np.float64, prefix='x', **kw); ref.train = False
ref([imgs, acts, stas], 0, tap_steps=steps)
m = pivp_amd.Model(10, prefix='x', keep_activations=True, **kw)
m.load_state_dict_reference(P)
with pivp_amd.using_config('train', False):
    m([imgs, acts, stas], 0)
gen = torch.stack(m.gen_images).cpu().numpy()


def theta_of(h5):
    s1 = R.relu(R.linear(h5.reshape(2, -1), P['model/stp_input/W'], P['model/stp_input/b']))
    return R.linear(s1, P['model/identity_params/W'], P['model/identity_params/b']) + np.array([[1.0, 0, 0, 0, 1.0, 0]])


for t in steps:
    h5 = m.tap('hidden5', t).cpu().numpy().astype(np.float64)
    th_hip = m.tap('stp_theta', t).cpu().numpy().astype(np.float64).reshape(2, 6)
    th_ref = theta_of(ref.taps[t]['hidden5'])
    th_mix = theta_of(h5)
    prev_ref = imgs[t].astype(np.float64) if t < 2 else ref.gen_images[t - 1]
    def warp(th):
        return R.spatial_transformer_sampler(prev_ref, R.spatial_transformer_grid(th.reshape(2, 2, 3), (64, 64)), 'clamp')
    w_ref = warp(th_ref)
    print('step %d: hidden5 rms %.2e | theta err: hip %.2e  f64-on-hip-h5 %.2e  (hip vs f64-on-hip-h5 %.2e)' % (
        t, np.sqrt(((h5 - ref.taps[t]['hidden5']) ** 2).mean()), np.abs(th_hip - th_ref).max(), np.abs(th_mix - th_ref).max(), np.abs(th_hip - th_mix).max()))
    print('   per-sample theta err (hip):', np.abs(th_hip - th_ref).max(axis=1))
    print('   warp-only output err with hip theta: %.2e ; total output err %.2e ; prev-frame err %.2e' % (
        R.per_pixel_l2(warp(th_hip), w_ref).max(), R.per_pixel_l2(gen[t], ref.gen_images[t]).max(),
        0.0 if t < 2 else R.per_pixel_l2(gen[t - 1], ref.gen_images[t - 1]).max()))
    # LayerNorm statistics the HIP path used, recovered from y = (h - mu) * rstd (gamma = 1, beta = 0 in this init)
    for nm, hid in (('lstm1', 'hidden1'), ('lstm5', 'hidden5')):
        h = m.tap(nm + '_h', t).cpu().numpy().astype(np.float64).reshape(2, -1)
        y = m.tap(hid, t).cpu().numpy().astype(np.float64).reshape(2, -1)
        for b in range(2):
            A = np.stack([h[b], np.ones_like(h[b])], 1)
            (a, c), *_ = np.linalg.lstsq(A, y[b], rcond=None)
            rstd = 1.0 / np.sqrt(h[b].var() + 1e-6)
            print('   %s sample %d: rstd rel err %.2e  mean err %.2e (std of h %.3f)' % (hid, b, a / rstd - 1, -c / a - h[b].mean(), h[b].std()))
