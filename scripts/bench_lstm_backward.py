"""Per-layer standalone timing of the fp32 ConvLSTM backward's two matrix kernels at config 2's shapes (B = 32 by default), random operands:
  * weight gradient: the round-2 kernel (atomics) against the round-6 kernel (partial slots; the once-per-sweep reduction timed separately),
  * data gradient (plain 5x5 conv of dG with the flipped, transposed weight),
  * the forward gate conv for reference.
Each timing: ~1 s of back-to-back launches first (the clock the chip HOLDS), then the mean of 20 launches by events.  With a stamp build
(PIVP_EXTRA_FLAGS='-DPIVP_WG_STAMPS -DPIVP_F32_STAMPS') it also prints the shader clock held inside the K loops (s_memtime / s_memrealtime) and the
blocks' phase times.    usage: bench_lstm_backward.py [B] [layers, e.g. lstm1,lstm7]"""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import pivp_amd  # noqa: F401
from pivp_amd import _lib

lib = _lib.load()
so = ctypes.CDLL(_lib.LIB_PATH)
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
only = sys.argv[2].split(',') if len(sys.argv) > 2 else None
st = torch.cuda.current_stream().cuda_stream
LAYERS = [('lstm1', 32, 32, 32), ('lstm3', 32, 64, 16), ('lstm4', 64, 64, 16), ('lstm5', 64, 128, 8), ('lstm6', 128, 64, 16), ('lstm7', 96, 32, 32)]
WARM = float(__import__('os').environ.get('WARM_S', '1.0'))


def timed(fn, reps=20):
    t0 = time.time()
    while time.time() - t0 < WARM:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def stamps(getter, nblk):
    if not hasattr(so, getter):
        return ''
    buf = (ctypes.c_longlong * (2048 * 8))()
    assert getattr(so, getter)(buf, 2048 * 8) == 0
    raw = np.array(list(buf), dtype=np.int64).reshape(2, 2048, 4)
    nblk = min(nblk, 2048)
    w = raw[0, :nblk].astype(np.float64) * 0.01; c = raw[1, :nblk].astype(np.float64)
    ok = (raw[0, :nblk, 2] > raw[0, :nblk, 1])
    if not ok.any():
        return ''
    ghz = (c[ok, 2] - c[ok, 1]) / np.maximum(raw[0, :nblk][ok, 2] - raw[0, :nblk][ok, 1], 1) * 0.1
    t0 = w[ok, 0].min()
    return ('   [stamps] clock in the K loop: median %.3f GHz (min %.3f max %.3f); entry spread %.1f us, prologue median %.1f, loop median %.1f / max %.1f, '
            'epilogue median %.1f, last end %.1f us' % (np.median(ghz), ghz.min(), ghz.max(), (w[ok, 0] - t0).max(), np.median(w[ok, 1] - w[ok, 0]),
                                                        np.median(w[ok, 2] - w[ok, 1]), (w[ok, 2] - w[ok, 1]).max(), np.median(w[ok, 3] - w[ok, 2]), (w[ok, 3] - t0).max()))


tot = {'wg_old': 0.0, 'wg_new': 0.0, 'dgrad': 0.0, 'fwd': 0.0, 'floor': 0.0}
for name, cx, C, H in LAYERS:
    mult = 2 if name == 'lstm1' else 1      # lstm2 has lstm1's shape
    if only and name not in only:
        continue
    cin, N, M = cx + C, 4 * C, B * H * H
    flop = 2.0 * 25 * cin * N * M
    floor = flop / 157.3e12 * 1e6
    x = torch.randn(B, H, H, cx, device=dev); h = torch.randn(B, H, H, C, device=dev) * 0.5; c = torch.randn(B, H, H, C, device=dev)
    dG = torch.randn(M, N, device=dev) * 0.1
    dW = torch.zeros(25 * cin * N, device=dev); db = torch.zeros(N, device=dev)
    w = torch.randn(25 * cin * N, device=dev) / np.sqrt(25 * cin); bias = torch.randn(N, device=dev) * 0.1
    nslot = max(lib.pivp_wgrad5x5_f32_part_floats(cx, C, B, H, H, 1), lib.pivp_wgrad5x5_f32_part_floats(cx, C, B, H, H, 2))
    part = torch.zeros(max(nslot, 1), device=dev)
    din = torch.empty(M, cin, device=dev); co = torch.empty_like(c); ho = torch.empty_like(h)

    def wg_old():
        assert lib.pivp_wgrad5x5_f32_batch(x.data_ptr(), cx, cx, h.data_ptr(), C, dG.data_ptr(), None, 0, dW.data_ptr(), db.data_ptr(), B, H, H, 1, 0, 0, 0, 0, st) == 0

    def wg_new(form):
        def f():
            assert lib.pivp_wgrad5x5_f32_batch(x.data_ptr(), cx, cx, h.data_ptr(), C, dG.data_ptr(), part.data_ptr(), 0, dW.data_ptr(), db.data_ptr(), B, H, H, 1, 0, 0, 0,
                                               form, st) == 0
        return f

    def wg_red(form):
        def f():
            assert lib.pivp_wgrad5x5_f32_reduce(cx, C, 1, part.data_ptr(), dW.data_ptr(), db.data_ptr(), B, H, H, form, st) == 0
        return f

    def dgrad():
        assert lib.pivp_conv5x5_f32(dG.data_ptr(), N, N, w.data_ptr(), din.data_ptr(), cin, B, H, H, st) == 0

    def fwd():
        assert lib.pivp_convlstm_v(x.data_ptr(), cx, cx, h.data_ptr(), C, w.data_ptr(), bias.data_ptr(), c.data_ptr(), co.data_ptr(), ho.data_ptr(), B, H, H, 0, st) == 0

    gates = torch.empty(M, N, device=dev)

    def fwd_train():
        assert lib.pivp_convlstm_train(x.data_ptr(), cx, cx, h.data_ptr(), C, w.data_ptr(), bias.data_ptr(), c.data_ptr(), co.data_ptr(), ho.data_ptr(), gates.data_ptr(), B, H, H, st) == 0

    print('%s: cin %d N %d map %d x %d B %d: %.2f GFLOP, floor %.1f us at 157.3 TF; partial slots %.1f MB' % (name, cin, N, H, H, B, flop / 1e9, floor, nslot * 4 / 1e6))
    t = timed(wg_old); tot['wg_old'] += mult * t
    print('   weight gradient, round-2 kernel (atomics)   %7.1f us  %.3f of peak' % (t, floor / t)); s = stamps('pivp_debug_wg_stamps', 1024); print(s) if s else None
    if nslot > 0:
        best = 1e9
        for form in (1, 2):
            t = timed(wg_new(form)); best = min(best, t)
            print('   weight gradient, round-6 kernel, %d columns  %7.1f us  %.3f of peak' % (32 * form, t, floor / t)); s = stamps('pivp_debug_wgp_stamps', 512); print(s) if s else None
            print('   ... its reduction (once per sweep)          %7.1f us' % timed(wg_red(form)))
        tot['wg_new'] += mult * best
    t = timed(dgrad); tot['dgrad'] += mult * t
    print('   data gradient                               %7.1f us  %.3f of peak' % (t, floor / t)); s = stamps('pivp_debug_f32_stamps', 512); print(s) if s else None
    t = timed(fwd); tot['fwd'] += mult * t
    print('   forward gate conv (inference)               %7.1f us  %.3f of peak' % (t, floor / t)); s = stamps('pivp_debug_f32_stamps', 512); print(s) if s else None
    t = timed(fwd_train); tot['fwd_train'] = tot.get('fwd_train', 0.0) + mult * t
    print('   forward gate conv (training: + gate activations) %5.1f us  %.3f of peak' % (t, floor / t)); s = stamps('pivp_debug_f32_stamps', 512); print(s) if s else None
    tot['floor'] += mult * floor
if not only:
    print('seven layers (lstm2 = lstm1): floor %.1f us; weight gradient round-2 %.1f (%.3f), round-6 %.1f (%.3f); data gradient %.1f (%.3f); forward %.1f (%.3f), in training %.1f (%.3f)' % (
        tot['floor'], tot['wg_old'], tot['floor'] / tot['wg_old'], tot['wg_new'], tot['floor'] / max(tot['wg_new'], 1e-9), tot['dgrad'], tot['floor'] / tot['dgrad'],
        tot['fwd'], tot['floor'] / tot['fwd'], tot.get('fwd_train', 1e-9), tot['floor'] / tot.get('fwd_train', 1e-9)))
