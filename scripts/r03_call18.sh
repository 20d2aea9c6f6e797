#!/bin/bash
# round 3, call 18: training plans also apply norm(hidden6 / hidden7) inside enc5 / enc6 (the kernel writes the normalised tensor and the
# statistics for the backward sweep): gradient tests, then the train step A/B (PIVP_LN_FOLD=2: inference only, 3: training too)
set -o pipefail
o=gpurun_out/r03/ln_in_deconv_train
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_trained.py tests/test_gpu_bf16.py tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_gpu_pipeline.py -x -q > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -2 $o/tests.txt
for v in 2 3 2 3; do
  PIVP_LN_FOLD=$v timeout -k 10 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $o/train_fp32_$v.json 2>$o/err.txt || { tail $o/err.txt; exit 1; }
  PIVP_LN_FOLD=$v timeout -k 10 300 python bench.py --mode train --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $o/train_bf16_$v.json 2>$o/err.txt || { tail $o/err.txt; exit 1; }
  python - <<PY
import json
a=json.loads(open('$o/train_fp32_$v.json').read().strip().splitlines()[-1]); b=json.loads(open('$o/train_bf16_$v.json').read().strip().splitlines()[-1])
print('PIVP_LN_FOLD=$v: train fp32 %.3f ms, bf16 %.3f ms' % (a['ms_per_step'], b['ms_per_step']), flush=True)
PY
done
