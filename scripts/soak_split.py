"""Soak of the split precision modes: the rollout is a pure function of its inputs (no atomics in the forward kernels), so N repetitions must give
bit-identical frames -- a missing barrier or an early fragment read shows up as a rare mismatch; then some train steps (finite, decreasing loss)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import pivp_amd
from oracle import restatement as R

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for size, B, T in ((64, 32, 10), (128, 4, 6)):
    rs = np.random.RandomState(3)
    imgs = rs.random_sample((T, B, 3, size, size)).astype(np.float32)
    acts = (0.1 * rs.standard_normal((T, B, 5))).astype(np.float32); stas = (0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)
    for prec in ('fp16x3', 'bf16x6'):
        m = pivp_amd.Model(10, prefix='s', precision=prec)
        with pivp_amd.using_config('train', False):
            m([imgs, acts, stas], 0)
            ref = torch.stack(m.gen_images).clone()
            bad = 0
            for i in range(N):
                m([imgs, acts, stas], 0)
                g = torch.stack(m.gen_images)
                if not torch.equal(g, ref):
                    bad += 1
        assert torch.isfinite(ref).all()
        print('%dx%d B=%d %s: %d rollouts, %d differ from the first' % (size, size, B, prec, N, bad), flush=True)
        assert bad == 0
    m = pivp_amd.Model(10, prefix='s', precision='fp16x3', keep_activations=True)
    opt = pivp_amd.Adam(alpha=1e-3).setup(m)
    losses = [float(opt.update(m, [imgs, acts, stas], 0)) for _ in range(25)]
    print('%dx%d fp16x3 train: loss %.5f -> %.5f over 25 steps' % (size, size, losses[0], losses[-1]), flush=True)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
