#!/bin/bash
# round 4, call 56: dG's maximum from the gate backward (an atomic maximum of float bits) instead of an absmax launch per cell and timestep
set -o pipefail
o=gpurun_out/r04/c56
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py tests/test_gpu_train.py tests/test_gpu_backward_ops.py -x -q -s -k "fp16x3 or split or refuse or gates" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
grep -h "two fp16 pieces max\|fp16x3 train step\|gradients" $o/tests.txt | cut -c1-250 | head -30
for rep in 1 2; do
for dg in 0 2 1; do
  PIVP_X3_DGRAD=$dg timeout -k 10 200 python bench.py --precision fp16x3 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train_dg$dg.json || exit 1
  python -c "import json; d=json.load(open('$o/train_dg$dg.json')); print('PIVP_X3_DGRAD=$dg train step', d['ms_per_step'])"
done
done
