#!/bin/bash
# round 4, call 35 (+ the enc5 / enc6 transposed convs as two fp16 pieces; was call 30): two fp16 pieces per operand, three MFMAs per product (precision 'fp16x3'): op tests, model tests, trained fixtures, layer times, bench
set -o pipefail
o=gpurun_out/r04/c35
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -s -k "fp16x3 or x6" > $o/tests_bf16.txt 2>&1 || { tail -40 $o/tests_bf16.txt; exit 1; }
grep -a "fp16 pieces\|fp16x3\|passed\|failed" $o/tests_bf16.txt | cut -c1-220
timeout -k 10 600 python -m pytest tests/test_gpu_trained.py -x -q -s -k "bf16x6 or fp16x3 or gradients" > $o/tests_trained.txt 2>&1 || { tail -40 $o/tests_trained.txt; exit 1; }
grep -a "rms ratio\|worst tensor\|passed\|failed" $o/tests_trained.txt | cut -c1-260
timeout -k 10 400 python bench.py --no-cpu-baseline > $o/bench.json 2> $o/bench.err && python - <<'EOF2'
import json
d = json.loads(open('gpurun_out/r04/c35/bench.json').read().strip().splitlines()[-1])
print('rollout', d['ms_per_step'])
for k in ('rollout_bf16x6', 'rollout_fp16x3'):
    print(k, d[k]['ms_per_step'], d[k].get('achieved_tflops'), d[k].get('per_layer_tflops'), d[k]['max_l2_vs_f32_rollout_per_step'][:3])
for k in ('train', 'train_bf16', 'train_bf16x6', 'train_fp16x3'):
    print(k, d[k]['ms_per_step'])
EOF2
