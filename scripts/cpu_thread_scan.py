import sys, os, time
sys.path.insert(0, '.')
import numpy as np, torch
sys.path.insert(0, 'scripts')
from cpu_baseline import rollout_rate
for B in (2, 32):
    for th in (1, 4, 8, 16, 32, 64, 128):      # (256 threads on the GPU box's CPU share: 0.1 frames/s at B = 2, minutes per call at B = 32)
        torch.set_num_threads(th)
        r, n = rollout_rate(B, 10, 6.0)
        print('B=%d threads=%d rollout %.1f frames/s (%d calls)' % (B, th, r, n), flush=True)
