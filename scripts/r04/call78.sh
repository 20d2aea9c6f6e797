#!/bin/bash
# round 4, call 78: the bf16x6 mode's ConvLSTM weight gradients with three bf16 pieces (25-tap kernel, two timesteps per launch) against the fp32 kernel
set -o pipefail
o=gpurun_out/r04/c78
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py tests/test_gpu_train.py -x -q -s -k "x6 or fp16x3 or split or refuse" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
grep -h "bf16x6 max\|bf16x6 train step\|bf16x6) gradients" $o/tests.txt | cut -c1-250 | head -30
for rep in 1 2; do
for wg in 0 1; do
  PIVP_X6_WGRAD=$wg timeout -k 10 200 python bench.py --precision bf16x6 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train_wg$wg.json || exit 1
  python -c "import json; d=json.load(open('$o/train_wg$wg.json')); print('PIVP_X6_WGRAD=$wg train step', d['ms_per_step'])"
done
done
