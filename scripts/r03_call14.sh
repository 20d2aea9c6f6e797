#!/bin/bash
# round 3, call 14: LayerNorm sums inside the gate backward, in the sweep: gradient tests, then the train step with / without it
set -o pipefail
mkdir -p gpurun_out/r03/ln_in_gates
o=gpurun_out/r03/ln_in_gates
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_trained.py tests/test_gpu_bf16.py tests/test_gpu_backward_ops.py -x -q > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -2 $o/tests.txt
for v in 0 1 0 1; do
  PIVP_LN_IN_GATES=$v timeout -k 10 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $o/train_fp32_$v.json 2>$o/err.txt || { tail $o/err.txt; exit 1; }
  PIVP_LN_IN_GATES=$v timeout -k 10 300 python bench.py --mode train --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $o/train_bf16_$v.json 2>$o/err.txt || { tail $o/err.txt; exit 1; }
  python - <<PY
import json
for k in ('fp32','bf16'):
    j=json.loads(open('$o/train_%s_$v.json'%k).read().strip().splitlines()[-1])
    print('PIVP_LN_IN_GATES=$v', k, 'train ms_per_step', j.get('ms_per_step'), flush=True)
PY
done
