"""How long does a small main-stream kernel take beside the side stream's ConvLSTM weight gradient?  Stream A runs lstm7's cell backward
(gate math + data gradient + wgrad5x5_kernel) in a loop; stream B times the enc5 data gradient's shape (conv3x3s2, 64 -> 96 channels on
a 32 x 32 map: igemm_small, 768 blocks) launch by launch with events: alone, and while A is busy."""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import pivp_amd  # noqa: F401
from pivp_amd import _lib

lib = _lib.load()
dev = 'cuda:0'
B = 32
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
# stream B's kernel: conv3x3s2 on (B, 32, 32, 64) -> (B, 16, 16, 96)
xc = torch.randn(B, 32, 32, 64, device=dev); wc = torch.randn(9 * 64 * 96, device=dev) * 0.05; oc = torch.empty(B, 16, 16, 96, device=dev)
# stream A: lstm7's backward (cx = 96, C = 32, 32 x 32)
cx, C, H = 96, 32, 32
x = torch.randn(B, H, H, cx, device=dev); h = torch.randn(B, H, H, C, device=dev) * 0.5
w = torch.randn(25 * (cx + C) * 4 * C, device=dev) / np.sqrt(25 * (cx + C))
gates = torch.rand(B, H, H, 4 * C, device=dev); c_old = torch.randn(B, H, H, C, device=dev); c_new = torch.randn(B, H, H, C, device=dev)
dh = torch.randn(B, H, H, C, device=dev); dc = torch.zeros(B, H, H, C, device=dev); dG = torch.empty(B, H, H, 4 * C, device=dev)
wt = torch.empty_like(w); d_in = torch.empty(B, H, H, cx + C, device=dev); dW = torch.zeros_like(w); db = torch.zeros(4 * C, device=dev)


def small(n):
    evs = []
    with torch.cuda.stream(sb):
        for _ in range(n):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.pivp_conv3x3s2(xc.data_ptr(), 64, 64, wc.data_ptr(), None, oc.data_ptr(), 96, 96, 0, B, 32, 32, sb.cuda_stream) == 0
            e1.record(); evs.append((e0, e1))
    return evs


def heavy(n):
    with torch.cuda.stream(sa):
        for _ in range(n):
            assert lib.pivp_convlstm_backward(x.data_ptr(), cx, cx, h.data_ptr(), C, w.data_ptr(), gates.data_ptr(), c_old.data_ptr(), c_new.data_ptr(),
                                              dh.data_ptr(), C, None, 0, dc.data_ptr(), 0, dG.data_ptr(), wt.data_ptr(), d_in.data_ptr(),
                                              dW.data_ptr(), db.data_ptr(), B, H, H, sa.cuda_stream) == 0


ye = torch.randn(B * 32 * 32 * 64, device=dev); ze = torch.empty_like(ye)


def elementwise(n):            # a kernel with no LDS and a handful of registers, same launch pattern
    evs = []
    with torch.cuda.stream(sb):
        for _ in range(n):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); torch.add(ye, 1.0, out=ze); e1.record(); evs.append((e0, e1))
    return evs


def report(name, ev, first=None):
    t = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
    extra = '' if first is None else '  (first %d launches: median %.1f)' % (first, np.median(t[:first]))
    print('%-34s median %6.1f us  p90 %6.1f  max %6.1f%s' % (name, np.median(t), np.percentile(t, 90), t.max(), extra))


elementwise(5); torch.cuda.synchronize()
ev = elementwise(50); torch.cuda.synchronize(); report('elementwise alone:', ev)
heavy(40); ev = elementwise(200); torch.cuda.synchronize(); report('elementwise beside cell backward:', ev, 100)
small(5); heavy(2); torch.cuda.synchronize()
ev = small(50); torch.cuda.synchronize()
t = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
print('small kernel alone:        median %.1f us  p90 %.1f  max %.1f' % (np.median(t), np.percentile(t, 90), t.max()))
heavy(40)                      # ~20 ms of work on stream A
ev = small(200); torch.cuda.synchronize()
t = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
print('beside the cell backward:  median %.1f us  p90 %.1f  max %.1f  (first 100 launches: median %.1f)' % (np.median(t), np.percentile(t, 90), t.max(), np.median(t[:100])))

# a trivial kernel with a chosen LDS request (scripts/micro/lds_probe.hip), 768 blocks like the enc5 data gradient
import ctypes, os
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'micro', 'liblds_probe.so')
if os.path.exists(so):
    pl = ctypes.CDLL(so)
    pl.lds_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    outb = torch.zeros(16, device=dev)

    def probe(n, lds, blocks, prio=0, work=500):
        evs = []
        with torch.cuda.stream(sb):
            for _ in range(n):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); assert pl.lds_probe_launch(outb.data_ptr(), lds, blocks, work, sb.cuda_stream, prio) == 0; e1.record(); evs.append((e0, e1))
        return evs
    for lds, blocks, prio, work in ((1024, 768, 0, 500), (1024, 768, 3, 500), (1024, 768, 0, 0), (1024, 768, 3, 0), (1024, 256, 0, 500), (1024, 256, 3, 500),
                                    (16384, 768, 0, 500), (16384, 768, 3, 500), (49152, 768, 0, 500), (49152, 768, 3, 500)):
        probe(3, lds, blocks, prio, work); torch.cuda.synchronize()
        ev = probe(30, lds, blocks, prio, work); torch.cuda.synchronize()
        ta = np.median([a.elapsed_time(b) * 1e3 for a, b in ev])
        heavy(20); ev = probe(60, lds, blocks, prio, work); torch.cuda.synchronize()
        tb = np.median([a.elapsed_time(b) * 1e3 for a, b in ev][:40])
        print('probe kernel, %6d B of LDS, %4d blocks, s_setprio %d, %3d dependent FMAs: alone %6.1f us, beside the cell backward %6.1f us' % (lds, blocks, prio, work, ta, tb))
