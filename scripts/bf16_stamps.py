"""In-kernel cycle stamps of the bf16 ConvLSTM kernel's phases (build with PIVP_EXTRA_FLAGS=-DPIVP_BF16_STAMPS):
entry | prologue done | first barrier passed | tap loop done | epilogue cells done, for wave 0 of block 37; the rest of the
launch's duration (hipEvents) is stores draining + the end of the grid."""
import os, sys, ctypes, glob, numpy as np, torch
sys.path.insert(0, '.')
import pivp_amd
from pivp_amd import _lib
X6 = os.environ.get('PIVP_STAMP_MODE', '1') == '6'      # the three-piece kernel (16-channel blocks, 16-wide maps)
lib = _lib.load()
so = ctypes.CDLL(_lib.LIB_PATH)
dev = 'cuda:0'; B = 32
st = torch.cuda.current_stream().cuda_stream
rs = np.random.RandomState(0)
for name, cx, C, H in [('lstm1', 32, 32, 32), ('lstm5', 64, 128, 8) if not X6 else ('lstm4', 64, 64, 16), ('lstm7', 96, 32, 32)]:
    x = torch.randn(B, H, H, cx, device=dev); h = torch.randn(B, H, H, C, device=dev) * 0.5; c = torch.randn(B, H, H, C, device=dev)
    w = torch.randn(25 * (cx + C) * 4 * C, device=dev) / np.sqrt(25 * (cx + C)); b = torch.randn(4 * C, device=dev) * 0.1
    co = torch.empty_like(c); ho = torch.empty_like(h)
    wb = torch.empty((3 if X6 else 1) * lib.pivp_lstm_bf16_weight_elems(cx + C, C), dtype=torch.int16, device=dev)
    assert (lib.pivp_pack_lstm_bf16x6 if X6 else lib.pivp_pack_lstm_bf16)(w.data_ptr(), wb.data_ptr(), cx + C, C, st) == 0
    def launch():
        if X6:
            assert lib.pivp_convlstm_bf16x6(x.data_ptr(), cx, cx, h.data_ptr(), C, wb.data_ptr(), b.data_ptr(), c.data_ptr(), co.data_ptr(),
                                            ho.data_ptr(), None, None, 0, None, B, H, H, 0, st) == 0
            return
        assert lib.pivp_convlstm_bf16(x.data_ptr(), cx, cx, h.data_ptr(), C, wb.data_ptr(), b.data_ptr(), c.data_ptr(), co.data_ptr(),
                                      ho.data_ptr(), None, None, 0, None, B, H, H, 0, st) == 0
    import time
    t_warm = time.time()                          # ~2 s of back-to-back launches first: the clock the chip HOLDS under this load, not a burst's
    while time.time() - t_warm < float(os.environ.get('STAMP_WARM_S', '2')):
        for _ in range(200):
            launch()
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record(); torch.cuda.synchronize()
    nch = 16 if X6 or (B * H * H // 128) * (C // 32) < 256 or C % 32 else 32
    nblk = min(2048, (B * H * H // 128) * (C // nch))
    buf = (ctypes.c_longlong * (2048 * 8))()
    assert so.pivp_debug_bf16_stamps(buf, 2048 * 8) == 0
    raw = np.array(list(buf), dtype=np.int64).reshape(2048, 8)[:nblk]
    v = raw * 0.01       # us (100 MHz counter)
    ghz = (raw[:, 7] - raw[:, 6]) / np.maximum(raw[:, 3] - raw[:, 2], 1) * 0.1       # shader cycles per 10 ns tick
    t0 = v[:, 0].min()

    def q(a):
        return 'min %.1f  median %.1f  p90 %.1f  max %.1f' % (a.min(), np.median(a), np.percentile(a, 90), a.max())
    print('%s: launch %.1f us by events, %d blocks' % (name, e0.elapsed_time(e1) / 20 * 1e3, nblk))
    print('   shader clock held inside the tap loop (s_memtime / s_memrealtime): median %.3f GHz, min %.3f, max %.3f' % (np.median(ghz), ghz.min(), ghz.max()))
    print('   entry            ', q(v[:, 0] - t0))
    print('   prologue         ', q(v[:, 1] - v[:, 0])); print('   first barrier    ', q(v[:, 2] - v[:, 1]))
    print('   tap loop         ', q(v[:, 3] - v[:, 2])); print('   gate cells       ', q(v[:, 4] - v[:, 3]))
    print('   stores drained   ', q(v[:, 5] - v[:, 4])); print('   end (from launch)', q(v[:, 5] - t0))
