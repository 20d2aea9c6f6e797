#!/bin/bash
# round 4, call 59: the fp16x3 mode's ConvLSTM weight gradients with two fp16 pieces (25-tap kernel, batched over timesteps) against the fp32 kernel
set -o pipefail
o=gpurun_out/r04/c59
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py tests/test_gpu_train.py -x -q -s -k "fp16x3 or split or refuse" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
grep -h "wgrad .* two fp16 pieces\|fp16x3 train step\|fp16x3) gradients" $o/tests.txt | cut -c1-250 | head -30
for rep in 1 2; do
for wg in 0 1; do
  PIVP_X3_WGRAD=$wg timeout -k 10 200 python bench.py --precision fp16x3 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train_wg$wg.json || exit 1
  python -c "import json; d=json.load(open('$o/train_wg$wg.json')); print('PIVP_X3_WGRAD=$wg train step', d['ms_per_step'])"
done
done
