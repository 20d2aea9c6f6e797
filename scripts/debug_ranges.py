import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import hip_ops as ops
import test_gpu_bf16 as T
x, h, c, W, b = [np.asarray(a, dtype=np.float32).astype(np.float64) for a in T._case(2, 32, 32, 16, 77)]
for xs, ws in ((1e3, 1e-3), (0.25, 4.0), (1e-2, 1e2), (1e-4, 1e4), (1.0, 1.0)):
    xx = np.asarray(x * xs, dtype=np.float32).astype(np.float64); WW = np.asarray(W * ws, dtype=np.float32).astype(np.float64)
    hr, cr, _ = T._lstm_ref(xx, h, c, WW, b)
    h3, c3 = ops.convlstm_fp16x3(xx, h, c, WW, b)
    h6, c6 = ops.convlstm_bf16x6(xx, h, c, WW, b)
    hf, cf = ops.convlstm(xx, h, c, WW, b)
    print(xs, ws, 'fp16x3 %.2e  bf16x6 %.2e  fp32 %.2e  nan in fp32: %d  max|pre-act| %.1f' % (np.abs(c3 - cr).max(), np.abs(c6 - cr).max(), np.nanmax(np.abs(cf - cr)), np.isnan(cf).sum(),
          np.abs(T.R.conv2d(np.concatenate([xx, h], 1), WW, b, 1, 2)).max()))
