#!/bin/bash
# round 4, call 69: do PIVP_X3_DGRAD / PIVP_X3_WGRAD reach the kernels they name?  One gradient per setting, compared pairwise
set -o pipefail
o=gpurun_out/r04/c69
mkdir -p $o
PIVP_X3_DGRAD=0 PIVP_X3_WGRAD=0 timeout -k 10 120 python scripts/r04/grad_dump.py $o/g00.npy 2>&1 | tail -1 || exit 1
PIVP_X3_DGRAD=1 PIVP_X3_WGRAD=0 timeout -k 10 120 python scripts/r04/grad_dump.py $o/g10.npy 2>&1 | tail -1 || exit 1
PIVP_X3_DGRAD=0 PIVP_X3_WGRAD=1 timeout -k 10 120 python scripts/r04/grad_dump.py $o/g01.npy 2>&1 | tail -1 || exit 1
timeout -k 10 120 python scripts/r04/grad_dump.py $o/g11.npy 2>&1 | tail -1 || exit 1
timeout -k 10 120 python scripts/r04/grad_dump.py $o/g11b.npy 2>&1 | tail -1 || exit 1
timeout -k 10 120 python scripts/r04/grad_dump.py $o/f32.npy fp32 2>&1 | tail -1 || exit 1
python - <<'EOF2'
import numpy as np
o = 'gpurun_out/r04/c69/'
g = {k: np.load(o + k + '.npy').astype(np.float64) for k in ('g00', 'g10', 'g01', 'g11', 'g11b', 'f32')}
rel = lambda a, b: np.linalg.norm(g[a] - g[b]) / np.linalg.norm(g[b])
print('default twice (atomics order):      %.3e' % rel('g11b', 'g11'))
print('fp16 data gradients vs bf16x6 ones: %.3e' % rel('g10', 'g00'))
print('fp16 weight gradients vs fp32 ones: %.3e' % rel('g01', 'g00'))
print('both vs neither:                    %.3e' % rel('g11', 'g00'))
for k in ('g00', 'g10', 'g01', 'g11'):
    print('%s vs the fp32 kernels: %.4e' % (k, rel(k, 'f32')))
EOF2
rm -f $o/*.npy
