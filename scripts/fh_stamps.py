"""Phase stamps of frame_head_kernel (build with PIVP_EXTRA_FLAGS=-DPIVP_FH_STAMPS): block (band 3, sample 5) of the last launch of a config-2
rollout, per wave, in microseconds from the block's first stamp.  Slots: 0 entry | 1 staging stored | 2 statistics merged | 3 past barrier (1) |
4 / 6 LayerNorm of pass 0 / 1 in LDS | 5 / 7 multiplies of pass 0 / 1 done | 8 the wave's tiles done (slots 4-7 of a wave with two tiles: its LAST tile) | 9 past barrier (2) |
10 group maxima done and past the finisher's barriers | 11 frame stored | 12 group loop done | 13 softmaxed masks in registers | 14 blended kernel in registers."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import pivp_amd
from pivp_amd import _lib

lib = _lib.load()
so = ctypes.CDLL(_lib.LIB_PATH)
rs = np.random.RandomState(0)
B, T, S = 32, 10, 64
images = torch.from_numpy(rs.random_sample((T, B, 3, S, S)).astype(np.float32)).cuda()
actions = torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).cuda()
states = torch.from_numpy((0.1 * rs.standard_normal((T, B, 5))).astype(np.float32)).cuda()
m = pivp_amd.Model(10, prefix='s', device='cuda:0')
with pivp_amd.using_config('train', False):
    for _ in range(5):
        m.reset_state(); m([images, actions, states], 0)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 128)()
assert so.pivp_debug_fh_stamps(buf) == 0
st = np.array(buf[:]).reshape(8, 16)
nw = int((st[:, 0] > 0).sum())
t0 = st[:nw, 0].min()
names = ['entry', 'staged', 'merged', 'bar1', 'ln0', 'mul0', 'ln1', 'mul1', 'tiles', 'bar2', 'groups', 'stored', 'gloop', 'masks', 'keff']
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 10, 13, 14, 11]
for w in range(nw):
    print('wave %d: ' % w + '  '.join('%s %.2f' % (names[i], (st[w, i] - t0) * 0.01) for i in order if st[w, i] > 0))
