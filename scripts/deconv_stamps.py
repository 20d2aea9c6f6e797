"""Per-block phase stamps of deconv3x3s2_tile_kernel<0, true> (build with PIVP_EXTRA_FLAGS=-DPIVP_DT_STAMPS; 100 MHz counter) at the rollout's enc6 / enc5 shapes,
B = 32: entry | LayerNorm statistics merged | first chunk staged (loop starts) | loop done | stores issued (+ statistics partial) | stores done."""
import ctypes, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import pivp_amd  # noqa: F401
from pivp_amd import _lib
lib = _lib.load(); so = ctypes.CDLL(_lib.LIB_PATH)
dev = 'cuda:0'; st = torch.cuda.current_stream().cuda_stream
B = 32
for name, c_ln, c1, cout, H in [('enc6', 32, 32, 64, 32), ('enc5', 64, 32, 96, 16)]:
    cin = c_ln + c1
    h = torch.randn(B, H, H, c_ln, device=dev); cat = torch.randn(B, H, H, cin, device=dev)
    g = torch.randn(H * H, c_ln, device=dev); be = torch.randn(H * H, c_ln, device=dev)
    w = torch.randn(9 * cin * cout, device=dev) / np.sqrt(9 * cin); bias = torch.randn(cout, device=dev)
    out = torch.empty(B, 2 * H, 2 * H, cout, device=dev)
    scratch = torch.empty(lib.pivp_layernorm_scratch_floats(B, c_ln * H * H), device=dev)
    def run():
        assert lib.pivp_deconv3x3s2_ln(h.data_ptr(), c_ln, cat.data_ptr() + c_ln * 4, c1, cin, w.data_ptr(), bias.data_ptr(), g.data_ptr(), be.data_ptr(), 1e-6,
                                       scratch.data_ptr(), out.data_ptr(), cout, cout, 1, B, H, H, 0, st) == 0
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(50): run()
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print('%s: B %d, %d+%d -> %d channels, %d x %d -> %d x %d: %.1f us per call (statistics launch + tile kernel)' % (name, B, c_ln, c1, cout, H, H, 2 * H, 2 * H, e0.elapsed_time(e1) / 20 * 1e3))
    if hasattr(so, 'pivp_debug_dt_stamps'):
        buf = (ctypes.c_longlong * (2048 * 8))()
        assert so.pivp_debug_dt_stamps(buf, 2048 * 8) == 0
        nblk = min(2048, B * (H // 8) * (H // 16) * (cout // 32))
        v = np.array(list(buf), dtype=np.int64).reshape(2048, 8)[:nblk, :6] * 0.01
        t0 = v[:, 0].min()
        q = lambda a: 'min %.1f median %.1f p90 %.1f max %.1f' % (a.min(), np.median(a), np.percentile(a, 90), a.max())
        print('   %d blocks; entry %s' % (nblk, q(v[:, 0] - t0)))
        for i, nm in enumerate(['statistics merge', 'first chunk (loads + LDS)', 'K loop', 'epilogue issue (+ partial)', 'store drain']):
            print('   %-28s %s' % (nm, q(v[:, i + 1] - v[:, i])))
        print('   block total %s; last end %.1f us' % (q(v[:, 5] - v[:, 0]), (v[:, 5] - t0).max()))
