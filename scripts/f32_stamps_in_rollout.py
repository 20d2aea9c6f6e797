"""Phase stamps of the LAST igemm_f32 launch of a config-2 rollout (lstm7 of the last timestep: build with PIVP_EXTRA_FLAGS=-DPIVP_F32_STAMPS), i.e. the
ConvLSTM kernel where the rollout runs it -- behind enc5's 20-us kernel, operands cold -- next to the same layer launched back to back (scripts/bench_lstm_backward.py)."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '.')
import pivp_amd
from pivp_amd import _lib
lib = _lib.load(); so = ctypes.CDLL(_lib.LIB_PATH)
rs = np.random.RandomState(0)
B, T, S = 32, 10, 64
images = torch.from_numpy(rs.random_sample((T, B, 3, S, S)).astype(np.float32)).cuda()
actions = torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).cuda()
states = torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).cuda()
m = pivp_amd.Model(10, prefix='s', device='cuda:0')
with pivp_amd.using_config('train', False):
    for _ in range(30):
        m.reset_state(); m([images, actions, states], 0)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (2048 * 8))()
assert so.pivp_debug_f32_stamps(buf, 2048 * 8) == 0
raw = np.array(list(buf), dtype=np.int64).reshape(2, 2048, 4)
nblk = 512
w = raw[0, :nblk].astype(np.float64) * 0.01; c = raw[1, :nblk].astype(np.float64)
ghz = (c[:, 2] - c[:, 1]) / np.maximum(raw[0, :nblk, 2] - raw[0, :nblk, 1], 1) * 0.1
t0 = w[:, 0].min()
q = lambda a: 'min %.1f median %.1f p90 %.1f max %.1f' % (a.min(), np.median(a), np.percentile(a, 90), a.max())
print('lstm7 inside the rollout (last timestep), %d blocks; clock in the loop median %.3f GHz' % (nblk, np.median(ghz)))
print('   entry    ', q(w[:, 0] - t0)); print('   prologue ', q(w[:, 1] - w[:, 0])); print('   loop     ', q(w[:, 2] - w[:, 1]))
print('   epilogue ', q(w[:, 3] - w[:, 2])); print('   end      ', q(w[:, 3] - t0))
