#!/bin/bash
# round 3, call 17: norms of hidden6 / hidden7 applied inside enc5 / enc6 (inference plans): op + model tests, then the rollout A/B
set -o pipefail
o=gpurun_out/r03/ln_in_deconv
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_trained.py tests/test_gpu_bf16.py -x -q > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -2 $o/tests.txt
for v in 1 2 1 2; do
  PIVP_LN_FOLD=$v timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train > $o/rollout_fold$v.json 2>$o/err.txt || { tail $o/err.txt; exit 1; }
  PIVP_LN_FOLD=$v timeout -k 10 300 python bench.py --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline --no-train --no-roofline > $o/rollout_bf16_fold$v.json 2>$o/err.txt || { tail $o/err.txt; exit 1; }
  python - <<PY
import json
a=json.loads(open('$o/rollout_fold$v.json').read().strip().splitlines()[-1]); b=json.loads(open('$o/rollout_bf16_fold$v.json').read().strip().splitlines()[-1])
print('PIVP_LN_FOLD=$v: rollout fp32 %.3f ms (conv frac %.4f), bf16 %.3f ms' % (a['ms_per_step'], a['roofline']['frac'], b['ms_per_step']), flush=True)
PY
done
