import sys, numpy as np, torch
sys.path.insert(0, '.')
import pivp_amd
from pivp_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
rs = np.random.RandomState(0)
for name, cx, C, H in [('lstm4', 64, 64, 16), ('lstm6', 128, 64, 16), ('lstm3', 32, 64, 16)]:
    B = 32
    x = torch.from_numpy(rs.randn(B, H, H, cx).astype(np.float32)).cuda(); h = torch.from_numpy((rs.randn(B, H, H, C) * .5).astype(np.float32)).cuda()
    c = torch.from_numpy(rs.randn(B, H, H, C).astype(np.float32)).cuda()
    w = torch.from_numpy((rs.randn(25 * (cx + C) * 4 * C) / np.sqrt(25 * (cx + C))).astype(np.float32)).cuda(); b = torch.from_numpy((rs.randn(4 * C) * .1).astype(np.float32)).cuda()
    outs = {}
    for v in (0, 4):
        co = torch.empty_like(c); ho = torch.empty_like(h)
        assert lib.pivp_convlstm_v(x.data_ptr(), cx, cx, h.data_ptr(), C, w.data_ptr(), b.data_ptr(), c.data_ptr(), co.data_ptr(), ho.data_ptr(), B, H, H, v, st) == 0
        torch.cuda.synchronize(); outs[v] = (co, ho)
    print(name, 'max |dc|', float((outs[0][0] - outs[4][0]).abs().max()), 'max |dh|', float((outs[0][1] - outs[4][1]).abs().max()))
