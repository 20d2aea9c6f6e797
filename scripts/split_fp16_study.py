"""CPU study (no GPU), companion of split_bf16_study.py: the gate convolutions' fp32 operands as TWO fp16 pieces (hi = fp16(v), lo = fp16(v - hi):
22 bits) and three products hi*hi + hi*lo + lo*hi with exact accumulation, the weights pre-scaled by a power of two so that their lo piece stays a
normal fp16 number.  Reports the per-pixel L2 against the unmodified float64 rollout of config 1, beside the three-bf16-piece emulation."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R


def f16(a):
    return np.asarray(a, dtype=np.float64).astype(np.float16).astype(np.float64)


def bf16(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).bfloat16().float().numpy().astype(np.float64)


def split(a, pieces, rnd):
    out, r = [], np.asarray(a, dtype=np.float32).astype(np.float64)
    for _ in range(pieces):
        p = rnd(r); out.append(p); r = r - p
    return out


def run(kind, wscale=256.0):
    orig = R.conv2d
    def conv(x, W, b=None, stride=1, pad=0):
        if kind == 'ref' or W.shape[2] != 5 or stride != 1:
            return orig(x, W, b, stride, pad)
        if kind == 'bf16x6':
            xs, ws, n, s = split(x, 3, bf16), split(W, 3, bf16), 3, 1.0
        else:
            xs, ws, n, s = split(x, 2, f16), split(W * wscale, 2, f16), 2, wscale
        y = 0.0
        for i in range(n):
            for j in range(n - i):
                y = y + orig(xs[i], ws[j], None, stride, pad)
        return y / s + (b.reshape(1, -1, 1, 1) if b is not None else 0.0)
    R.conv2d = conv
    try:
        P = R.init_params(seed=1, dtype=np.float32, scale=1.0)
        imgs, acts, stas = R.synthetic_batch(2, 10)
        m = R.Model(10, params=P, dtype=np.float64, prefix='s'); m.train = False
        m([imgs, acts, stas], 0)
        return np.stack(m.gen_images)
    finally:
        R.conv2d = orig


if __name__ == '__main__':
    torch.set_num_threads(8)
    t0 = time.time(); ref = run('ref'); print('reference rollout %.0f s' % (time.time() - t0), flush=True)
    for kind, ws in (('bf16x6', 1.0), ('fp16x3', 256.0), ('fp16x3', 1.0)):
        g = run(kind, ws)
        l2 = R.per_pixel_l2(g, ref)
        print('%s (weights x %g): per-pixel L2 max %.2e rms %.2e; first frame max %.2e, last frame max %.2e' %
              (kind, ws, l2.max(), np.sqrt((l2 ** 2).mean()), l2[0].max(), l2[-1].max()), flush=True)
