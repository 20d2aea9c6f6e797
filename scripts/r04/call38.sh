#!/bin/bash
# round 4, call 38: fp16 pieces on 8-wide maps (lstm5 at 64 x 64 frames) through the ring kernel: op tests, trained fixtures, layer times, bench
set -o pipefail
o=gpurun_out/r04/c38
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py -x -q -s -k "fp16x3" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
grep -a "H=8\|rms ratio\|worst tensor\|passed\|failed" $o/tests.txt | cut -c1-260
PIVP_BENCH_INTERLEAVE=1 PIVP_BENCH_BF16=h3 timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 2>&1 | grep -v amdgpu.ids | tee $o/layers_h3.txt
timeout -k 10 400 python bench.py --no-cpu-baseline > $o/bench.json 2> $o/bench.err && python - <<'EOF2'
import json
d = json.loads(open('gpurun_out/r04/c38/bench.json').read().strip().splitlines()[-1])
print('rollout', d['ms_per_step'], {k: d[k]['ms_per_step'] for k in d if k.startswith('rollout_') or k.startswith('train')}, d['rollout_fp16x3'].get('per_layer_tflops'))
EOF2
