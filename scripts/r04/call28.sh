#!/bin/bash
# round 4, call 28 (data gradient on the eight-wave L2-direct kernel; was call 25): the three-piece data gradient (conv5x5 with six MFMAs per product on the k-step ring): op tests, train-step gradient tests, train step A/B
set -o pipefail
o=gpurun_out/r04/c28
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -s -k "x6" > $o/tests_bf16.txt 2>&1 || { tail -40 $o/tests_bf16.txt; exit 1; }
grep -a "conv5x5\|bf16x6\|passed\|failed" $o/tests_bf16.txt | cut -c1-220
timeout -k 10 300 python bench.py --no-cpu-baseline > $o/bench.json 2> $o/bench.err && python - <<'EOF2'
import json
d = json.loads(open('gpurun_out/r04/c28/bench.json').read().strip().splitlines()[-1])
print('rollout', d['ms_per_step'], 'x6', d['rollout_bf16x6']['ms_per_step'], d['rollout_bf16x6']['max_l2_vs_f32_rollout_per_step'])
for k in ('train', 'train_bf16', 'train_bf16x6'):
    print(k, d[k]['ms_per_step'], d[k]['loss'])
EOF2
