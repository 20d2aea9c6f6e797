#!/bin/bash
# round 4, call 26 (the 4 x 2-wave form of the 16-channel blocks; was call 24): three-piece kernel with the weights from L2 into registers (32-channel blocks): op tests, trained fixtures, per-layer times, rollout
set -o pipefail
o=gpurun_out/r04/c26
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -s -k "x6" > $o/tests_bf16.txt 2>&1 || { tail -40 $o/tests_bf16.txt; exit 1; }
grep -a "three-piece\|bf16x6\|passed\|failed" $o/tests_bf16.txt | cut -c1-220
timeout -k 10 600 python -m pytest tests/test_gpu_trained.py -x -q -s -k "bf16x6" > $o/tests_trained.txt 2>&1 || { tail -40 $o/tests_trained.txt; exit 1; }
grep -a "rms ratio\|passed\|failed" $o/tests_trained.txt | cut -c1-260
for v in 1 2 16 0; do
  echo "== nch $v" | tee -a $o/layers_x6.txt
  PIVP_LSTM_VARIANT=$v PIVP_BENCH_INTERLEAVE=1 PIVP_BENCH_BF16=6 timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 2>&1 | grep -v amdgpu.ids | tee -a $o/layers_x6.txt || exit 1
done
timeout -k 10 300 python bench.py --no-cpu-baseline > $o/bench.json 2> $o/bench.err && python - <<'EOF2'
import json
d = json.loads(open('gpurun_out/r04/c26/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], json.dumps(d.get('rollout_bf16x6')))
for k in ('train', 'train_bf16', 'train_bf16x6'):
    print(k, d[k]['ms_per_step'])
EOF2
