#!/bin/bash
# round 3, call 26: config 5's per-GPU share (B = 32, T = 20, 128 x 128): rollout and train step with / without the norms folded into enc5 / enc6
set -o pipefail
o=gpurun_out/r03/config5
mkdir -p $o
for v in 1 3 4; do
  PIVP_LN_FOLD=$v timeout -k 10 400 python bench.py --size 128 --seq-len 20 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $o/bench_fold$v.json 2>$o/err.txt || { tail $o/err.txt; exit 1; }
  python - <<PY
import json
d=json.loads(open('$o/bench_fold$v.json').read().strip().splitlines()[-1])
print('PIVP_LN_FOLD=$v: 128x128 T=20 rollout %.2f ms, train %.2f ms, train_bf16 %.2f ms' % (d['ms_per_step'], d['train']['ms_per_step'], d['train_bf16']['ms_per_step']), flush=True)
PY
done
