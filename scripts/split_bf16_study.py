"""CPU study (no GPU): what would the ConvLSTM gate convolutions cost in accuracy if their fp32 operands were split into bf16 pieces
(x = x_hi + x_lo [+ x_lo2]) and multiplied piece by piece with exact accumulation -- the "split-bf16" emulation named in DESIGN.md 8?
Runs the float64 restatement on config 1 (B = 2, T = 10, CDNA) with the gate conv replaced by the emulation and reports the
per-pixel L2 against the unmodified float64 rollout.  pieces = 2: hi*hi + hi*lo + lo*hi (3 MFMAs); 3: the 6 products down to 2^-24."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R


def bf16(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).bfloat16().float().numpy().astype(np.float64)


def split(a, pieces):
    out, r = [], np.asarray(a, dtype=np.float32).astype(np.float64)   # the kernel's operands are fp32
    for _ in range(pieces):
        p = bf16(r); out.append(p); r = r - p
    return out


def run(pieces):
    orig = R.conv2d
    def conv(x, W, b=None, stride=1, pad=0):
        if pieces == 0 or W.shape[2] != 5 or stride != 1:      # only the 5x5 stride-1 gate convolutions
            return orig(x, W, b, stride, pad)
        xs, ws = split(x, pieces), split(W, pieces)
        y = 0.0
        for i in range(pieces):
            for j in range(pieces - i):                         # drop the products below the last kept order
                y = y + orig(xs[i], ws[j], None, stride, pad)
        return y + (b.reshape(1, -1, 1, 1) if b is not None else 0.0)
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
    t0 = time.time(); ref = run(0); print('reference rollout %.0f s' % (time.time() - t0), flush=True)
    for pieces in (1, 2, 3):
        g = run(pieces)
        l2 = R.per_pixel_l2(g, ref)
        print('pieces %d (%d bf16 MFMAs per product): per-pixel L2 max %.2e rms %.2e; last frame max %.2e' %
              (pieces, pieces * (pieces + 1) // 2, l2.max(), np.sqrt((l2 ** 2).mean()), l2[-1].max()), flush=True)
