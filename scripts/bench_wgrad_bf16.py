"""ConvLSTM weight gradient in the bf16 mode, per layer at B = 32 (config 3's per-GPU batch): microseconds and TFLOP/s per launch for one
timestep and for a batch of timesteps per launch (pivp_wgrad5x5_bf16_batch):   python3 scripts/bench_wgrad_bf16.py [B] [tcount ...]"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import pivp_amd  # noqa: F401
from pivp_amd import _lib

lib = _lib.load()
dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tcs = [int(a) for a in sys.argv[2:]] or [1, 8]
st = torch.cuda.current_stream().cuda_stream
LAYERS = [('lstm1', 32, 32, 32), ('lstm3', 32, 64, 16), ('lstm4', 64, 64, 16), ('lstm5', 64, 128, 8), ('lstm6', 128, 64, 16), ('lstm7', 96, 32, 32)]
tot = {tc: [0.0, 0.0] for tc in tcs}
for name, cx, C, H in LAYERS:
    for tc in tcs:
        x = torch.randn(tc, B, H, H, cx, device=dev); h = torch.randn(tc, B, H, H, C, device=dev) * 0.5
        g = torch.randn(tc, B, H, H, 4 * C, device=dev) * 0.1
        dW = torch.zeros(25 * (cx + C) * 4 * C, device=dev); db = torch.zeros(4 * C, device=dev)
        sx, sh, sg = x[0].numel() * 4, h[0].numel() * 4, g[0].numel() * 4

        def launch():
            rc = lib.pivp_wgrad5x5_bf16_batch(x[tc - 1].data_ptr(), cx, cx, h[tc - 1].data_ptr(), C, g.data_ptr(), dW.data_ptr(), db.data_ptr(),
                                              B, H, H, tc, -sx, -sh, sg, st)
            assert rc == 0, rc
        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n):
            launch()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        flop = 2.0 * tc * B * H * H * 25 * (cx + C) * 4 * C
        w = 2 if name == 'lstm1' else 1                        # lstm2 has lstm1's shape
        tot[tc][0] += w * us / tc; tot[tc][1] += w * flop / tc
        print('%-6s timesteps per launch %d: %8.1f us per launch = %7.1f us per timestep, %7.1f TFLOP/s = %.3f of the 2.5 PF bf16 peak' % (
            name, tc, us, us / tc, flop / us * 1e-6, flop / us * 1e-6 / 2500.0))
for tc in tcs:
    print('all seven layers, %d timestep(s) per launch: %.1f us per timestep, %.1f TFLOP/s = %.3f of peak' % (
        tc, tot[tc][0], tot[tc][1] / tot[tc][0] * 1e-6, tot[tc][1] / tot[tc][0] * 1e-6 / 2500.0))
