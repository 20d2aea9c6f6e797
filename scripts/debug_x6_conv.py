"""Error distribution of the three-piece 5x5 convolution (K split over channel groups) against float64."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import hip_ops as ops
from oracle import restatement as R
for B, cin, cout, H in [(32, 256, 192, 16), (8, 256, 192, 16), (32, 256, 64, 16), (32, 128, 192, 16)]:
    rs = np.random.RandomState(1)
    x = rs.randn(B, cin, H, H).astype(np.float32).astype(np.float64); W = (rs.randn(cout, cin, 5, 5) / np.sqrt(25 * cin)).astype(np.float32).astype(np.float64)
    ref = R.conv2d(x, W, np.zeros(cout), 1, 2)
    o1 = ops.conv5x5_bf16(x, W, pieces=3); o2 = ops.conv5x5_bf16(x, W, pieces=3)
    e = np.abs(o1 - ref)
    i = np.unravel_index(e.argmax(), e.shape)
    print(B, cin, cout, H, 'max %.2e rms %.2e p99.99 %.2e at' % (e.max(), np.sqrt((e ** 2).mean()), np.percentile(e, 99.99)), i, 'ref there %.3f' % ref[i],
          'repeatable' if np.array_equal(o1, o2) else 'differs between calls by %.2e' % np.abs(o1 - o2).max(),
          'count > 2e-6: %d' % (e > 2e-6).sum(), 'channels of those:', np.unique(np.where(e > 2e-6)[1])[:20], 'rows', np.unique(np.where(e > 2e-6)[2])[:20])
