#!/bin/bash
# round 4, call 67: bf16 train step with the 25-tap kernel on three quarters of the CUs (default) against every CU
set -o pipefail
o=gpurun_out/r04/c67
mkdir -p $o
for rep in 1 2 3; do
  timeout -k 10 200 python bench.py --precision bf16 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train.json || exit 1
  python -c "import json; d=json.load(open('$o/train.json')); print('default (25-tap kernel: 192 blocks) train step', d['ms_per_step'])"
  PIVP_WGB_SLOTS=192 timeout -k 10 200 python bench.py --precision bf16 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train.json || exit 1
  python -c "import json; d=json.load(open('$o/train.json')); print('PIVP_WGB_SLOTS=192 (both kernels) train step', d['ms_per_step'])"
done
timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train.py -x -q -k "bf16 and not x6 and not fp16x3" 2>&1 | tail -1
