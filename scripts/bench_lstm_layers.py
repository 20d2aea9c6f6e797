"""Per-layer microbenchmark of the ConvLSTM kernel at the B=32 shapes of config 2 (hipEvent timing,
interleaved rounds in one process).  Also the target for rocprofv3 --pmc passes."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import pivp_amd
from pivp_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
only = sys.argv[3].split(',') if len(sys.argv) > 3 else None
LAYERS = [('lstm1', 32, 32, 32), ('lstm2', 32, 32, 32), ('lstm3', 32, 64, 16), ('lstm4', 64, 64, 16),
          ('lstm5', 64, 128, 8), ('lstm6', 128, 64, 16), ('lstm7', 96, 32, 32)]
import os
VARIANT = int(os.environ.get('PIVP_LSTM_VARIANT', '0'))
BF16 = os.environ.get('PIVP_BENCH_BF16', '0') in ('1', '3', '6', 'h3')   # bf16-operand kernel (VARIANT = channels per block: 0 / 16 / 32); 3 = split mode; 6 = three pieces
X3 = os.environ.get('PIVP_BENCH_BF16', '0') == '3'
X6 = os.environ.get('PIVP_BENCH_BF16', '0') == '6'
H3 = os.environ.get('PIVP_BENCH_BF16', '0') == 'h3'      # two fp16 pieces, three MFMAs per product
DATA = os.environ.get('PIVP_BENCH_DATA', 'random')   # random | zero | const: does the MFMA rate depend on the operand values?
lib = _lib.load()
dev = 'cuda:0'
st = torch.cuda.current_stream().cuda_stream
rs = np.random.RandomState(0)
bufs = {}
for name, cx, C, H in LAYERS:
    if only and name not in only:
        continue
    x = torch.from_numpy(rs.randn(B, H, H, cx).astype(np.float32)).to(dev)
    h = torch.from_numpy((rs.randn(B, H, H, C) * 0.5).astype(np.float32)).to(dev)
    c = torch.from_numpy(rs.randn(B, H, H, C).astype(np.float32)).to(dev)
    w = torch.from_numpy((rs.randn(25 * (cx + C) * 4 * C) / np.sqrt(25 * (cx + C))).astype(np.float32)).to(dev)
    b = torch.from_numpy((rs.randn(4 * C) * 0.1).astype(np.float32)).to(dev)
    if DATA == 'zero':
        x.zero_(); h.zero_(); w.zero_()
    elif DATA == 'const':
        x.fill_(0.37); h.fill_(-0.21); w.fill_(0.013)
    co = torch.empty_like(c); ho = torch.empty_like(h)
    wb = None
    if BF16:
        if X6 and H % 16:
            continue                      # (8-wide maps are not the three-piece kernel's)
        wb = torch.empty((3 if X6 else 2 if (X3 or H3) else 1) * lib.pivp_lstm_bf16_weight_elems(cx + C, C) + 256, dtype=torch.int16, device=dev)
        if H3:
            assert lib.pivp_pack_lstm_fp16x3(w.data_ptr(), wb.data_ptr(), cx + C, C, H, st) == 0
        else:
            assert (lib.pivp_pack_lstm_bf16x6 if X6 else lib.pivp_pack_lstm_bf16x3 if X3 else lib.pivp_pack_lstm_bf16)(w.data_ptr(), wb.data_ptr(), cx + C, C, st) == 0
    bufs[name] = (x, h, c, w, b, co, ho, cx, C, H, wb)

def launch(name):
    x, h, c, w, b, co, ho, cx, C, H, wb = bufs[name]
    if X6 or H3:
        rc = (lib.pivp_convlstm_fp16x3 if H3 else lib.pivp_convlstm_bf16x6)(x.data_ptr(), cx, cx, h.data_ptr(), C, wb.data_ptr(), b.data_ptr(), c.data_ptr(), co.data_ptr(),
                                      ho.data_ptr(), None, None, 0, None, B, H, H, VARIANT, st)
        assert rc == 0, rc
        return
    if X3:
        rc = lib.pivp_convlstm_bf16x3(x.data_ptr(), cx, cx, h.data_ptr(), C, wb.data_ptr(), b.data_ptr(), c.data_ptr(), co.data_ptr(),
                                      ho.data_ptr(), None, None, 0, None, B, H, H, VARIANT, st)
        assert rc == 0, rc
        return
    if BF16:
        rc = lib.pivp_convlstm_bf16(x.data_ptr(), cx, cx, h.data_ptr(), C, wb.data_ptr(), b.data_ptr(), c.data_ptr(), co.data_ptr(),
                                    ho.data_ptr(), None, None, 0, None, B, H, H, VARIANT, st)
        assert rc == 0, rc
        return
    rc = lib.pivp_convlstm_v(x.data_ptr(), cx, cx, h.data_ptr(), C, w.data_ptr(), b.data_ptr(), c.data_ptr(), co.data_ptr(),
                             ho.data_ptr(), B, H, H, VARIANT, st)
    assert rc == 0, rc

for name in bufs:
    launch(name)
torch.cuda.synchronize()
res = {n: [] for n in bufs}
if os.environ.get('PIVP_BENCH_INTERLEAVE', '0') == '1':
    # the layers in program order, one launch each per pass (as in a rollout: every launch finds the caches as the other six left them)
    for r in range(5):
        evs = []
        for _ in range(iters):
            for name in bufs:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); launch(name); e1.record()
                evs.append((name, e0, e1))
        torch.cuda.synchronize()
        acc = {n: 0.0 for n in bufs}
        for name, e0, e1 in evs:
            acc[name] += e0.elapsed_time(e1)
        for name in bufs:
            res[name].append(acc[name] / iters)
else:
    for r in range(5):
        for name in bufs:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                launch(name)
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / iters)
tot_f, tot_t = 0.0, 0.0
for name in bufs:
    x, h, c, w, b, co, ho, cx, C, H, wb = bufs[name]
    fl = 2.0 * B * H * H * 4 * C * 25 * (cx + C)
    ms = float(np.median(res[name]))
    tot_f += fl; tot_t += ms
    print('%s  M=%6d N=%4d K=%5d  %8.1f us  %6.1f TFLOP/s  (min %.1f us)' % (name, B * H * H, 4 * C, 25 * (cx + C), ms * 1e3, fl / ms / 1e9, min(res[name]) * 1e3))
print('sum  %8.1f us  %6.1f TFLOP/s' % (tot_t * 1e3, tot_f / tot_t / 1e9))
