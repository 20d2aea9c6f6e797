"""CPU study (float32 NumPy oracle vs the float64 oracle): how much of the hidden5 / STP output error comes from the gate
epilogue's transcendental formulas.  Variants: 'exact' = numpy tanh; 'fast' = 2/(1+exp(-2x))-1 and 1/(1+exp(-x)) in float32
(the round-1 HIP epilogue); 'acc' = the round-2 epilogue (odd polynomial for |x| < 0.25 ... see igemm_f32.hip)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle import restatement as R

f32 = np.float32


def fast_tanh(x):
    return (f32(2.0) / (f32(1.0) + np.exp(f32(-2.0) * x)) - f32(1.0)).astype(f32)


def fast_sigmoid(x):
    return (f32(1.0) / (f32(1.0) + np.exp(-x))).astype(f32)


def acc_tanh(x):
    # tanh(x) = -em1/(2+em1), em1 = expm1(-2|x|): accurate near 0
    a = np.abs(x)
    em1 = np.expm1(f32(-2.0) * a).astype(f32)
    t = (-em1 / (f32(2.0) + em1)).astype(f32)
    return np.copysign(t, x).astype(f32)


class M(R.Model):
    variant = 'exact'

    def _lstm(self, name, inputs, forget_bias=1.0):
        if self.variant == 'exact' or self.dtype != np.float32:
            return R.Model._lstm(self, name, inputs, forget_bias)
        C = R.LSTM_SIZES[name]
        B, _, H, W = inputs.shape
        if self.lstm_c[name] is None:
            self.lstm_c[name] = np.zeros((B, C, H, W), dtype=self.dtype)
            self.lstm_h[name] = np.zeros((B, C, H, W), dtype=self.dtype)
        x = np.concatenate((inputs, self.lstm_h[name]), axis=1)
        g = R.conv2d(x, self.p[name + '/conv/W'], self.p[name + '/conv/b'], 1, 2)
        j, i, f, o = np.split(g, 4, axis=1)
        th, sg = (fast_tanh, fast_sigmoid) if self.variant == 'fast' else (acc_tanh, fast_sigmoid)
        c = self.lstm_c[name] * sg(f + f32(forget_bias)) + sg(i) * th(j)
        h = th(c) * sg(o)
        self.lstm_c[name], self.lstm_h[name] = c.astype(f32), h.astype(f32)
        return self.lstm_h[name]


T = 4
P = R.init_params(seed=1, dtype=np.float64, scale=1.0, num_masks=10, model_type='STP')
imgs, acts, stas = R.synthetic_batch(2, T)
steps = tuple(range(T - 1))
ref = M(10, is_cdna=False, is_stp=True, params=P, dtype=np.float64, prefix='x'); ref.train = False
ref([imgs, acts, stas], 0, tap_steps=steps)
for variant in ('exact', 'fast', 'acc'):
    m = M(10, is_cdna=False, is_stp=True, params=P, dtype=np.float32, prefix='x'); m.train = False
    m.variant = variant
    m([imgs, acts, stas], 0, tap_steps=steps)
    for t in steps:
        e5 = m.taps[t]['hidden5'] - ref.taps[t]['hidden5']
        e1 = m.taps[t]['hidden1'] - ref.taps[t]['hidden1']
        l2 = R.per_pixel_l2(m.gen_images[t], ref.gen_images[t])
        print('%-6s step %d: hidden1 rms %.2e  hidden5 rms %.2e max %.2e | output max %.2e rms %.2e' % (
            variant, t, np.sqrt((e1 ** 2).mean()), np.sqrt((e5 ** 2).mean()), np.abs(e5).max(), l2.max(), np.sqrt((l2 ** 2).mean())))
