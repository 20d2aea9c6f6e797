#!/bin/bash
# round 4, call 40: two k-steps per wait in the eight-wave two-fp16-piece kernels (default) against one (variant build): tests, layer times, bench
set -o pipefail
o=gpurun_out/r04/c40
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py -x -q -k "fp16x3" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
for lib in physical-interaction-video-prediction_amd/libpivp_hip.so physical-interaction-video-prediction_amd/variants/libpivp_hip_single.so; do
  echo "== $lib" | tee -a $o/layers.txt
  PIVP_BENCH_LIB=$lib PIVP_BENCH_INTERLEAVE=1 PIVP_BENCH_BF16=h3 timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 2>&1 | grep -v amdgpu.ids | tee -a $o/layers.txt || exit 1
done
timeout -k 10 400 python bench.py --no-cpu-baseline --no-train > $o/bench.json 2> $o/bench.err && python - <<'EOF2'
import json
d = json.loads(open('gpurun_out/r04/c40/bench.json').read().strip().splitlines()[-1])
print('rollout', d['ms_per_step'], {k: d[k]['ms_per_step'] for k in d if k.startswith('rollout_')})
EOF2
