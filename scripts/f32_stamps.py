"""Per-block phase stamps of the fp32 ConvLSTM kernel (build with PIVP_EXTRA_FLAGS=-DPIVP_F32_STAMPS; 100 MHz counter, 10 ns):
for every block entry | first chunk staged (loop starts) | loop done | stores done.  Prints, per layer, the launch's duration by
hipEvents and the distribution over blocks of: entry time (dispatch ramp), prologue, loop, epilogue, and the end time (tail)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import pivp_amd  # noqa: F401
from pivp_amd import _lib

lib = _lib.load()
so = ctypes.CDLL(_lib.LIB_PATH)
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
st = torch.cuda.current_stream().cuda_stream
for name, cx, C, H in [('lstm1', 32, 32, 32), ('lstm4', 64, 64, 16), ('lstm5', 64, 128, 8), ('lstm6', 128, 64, 16), ('lstm7', 96, 32, 32)]:
    x = torch.randn(B, H, H, cx, device=dev); h = torch.randn(B, H, H, C, device=dev) * 0.5; c = torch.randn(B, H, H, C, device=dev)
    w = torch.randn(25 * (cx + C) * 4 * C, device=dev) / np.sqrt(25 * (cx + C)); b = torch.randn(4 * C, device=dev) * 0.1
    co = torch.empty_like(c); ho = torch.empty_like(h)

    def launch():
        assert lib.pivp_convlstm_v(x.data_ptr(), cx, cx, h.data_ptr(), C, w.data_ptr(), b.data_ptr(), c.data_ptr(), co.data_ptr(),
                                   ho.data_ptr(), B, H, H, 0, st) == 0
    import time
    t_warm = time.time()                          # ~2 s of back-to-back launches first: the clock the chip HOLDS under this load, not a burst's
    while time.time() - t_warm < float(__import__('os').environ.get('STAMP_WARM_S', '2')):
        for _ in range(200):
            launch()
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record(); torch.cuda.synchronize()
    nblk = min(2048, (B * H * H // 64) * (C // 32))
    if nblk < 256:
        nblk = min(2048, (B * H * H // 32) * (C // 32))
    buf = (ctypes.c_longlong * (2048 * 8))()
    assert so.pivp_debug_f32_stamps(buf, 2048 * 8) == 0
    raw = np.array(list(buf), dtype=np.int64).reshape(2, 2048, 4)
    v = raw[0, :nblk] * 0.01     # us
    cyc = raw[1, :nblk].astype(np.float64)
    ghz = (cyc[:, 2] - cyc[:, 1]) / np.maximum(raw[0, :nblk, 2] - raw[0, :nblk, 1], 1) * 0.1      # cycles per 10 ns tick -> GHz
    print('   shader clock held inside the K loop (s_memtime / s_memrealtime): median %.3f GHz, min %.3f, max %.3f' % (np.median(ghz), ghz.min(), ghz.max()))
    t0 = v[:, 0].min()
    ent, pro, loop, epi, end = v[:, 0] - t0, v[:, 1] - v[:, 0], v[:, 2] - v[:, 1], v[:, 3] - v[:, 2], v[:, 3] - t0

    def q(a):
        return 'min %.1f  median %.1f  p90 %.1f  max %.1f' % (a.min(), np.median(a), np.percentile(a, 90), a.max())
    print('%s: launch %.1f us by events, %d blocks; first entry -> last store %.1f us' % (name, e0.elapsed_time(e1) / 20 * 1e3, nblk, end.max()))
    print('   entry    ', q(ent)); print('   prologue ', q(pro)); print('   loop     ', q(loop)); print('   epilogue ', q(epi)); print('   end      ', q(end))
    # workgroups are dealt round-robin to the 8 XCDs: block b runs on XCD b % 8
    xcd = np.arange(nblk) % 8
    print('   per XCD: loop median ', ' '.join('%.1f' % np.median(loop[xcd == k]) for k in range(8)), '| end max ', ' '.join('%.1f' % end[xcd == k].max() for k in range(8)))
    order = np.argsort(end)
    print('   last 8 blocks to end:', [(int(i), int(i) % 8, round(float(loop[i]), 1), round(float(epi[i]), 1)) for i in order[-8:]])
    print('   first 8 blocks to end:', [(int(i), int(i) % 8, round(float(loop[i]), 1), round(float(epi[i]), 1)) for i in order[:8]])
