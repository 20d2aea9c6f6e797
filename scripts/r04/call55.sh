#!/bin/bash
# round 4, call 55: the fp16x3 mode's data gradients with two fp16 pieces (dG times a power of two from its largest value) against the three-bf16-piece form
set -o pipefail
o=gpurun_out/r04/c55
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py tests/test_gpu_train.py -x -q -k "fp16x3 or split or refuse" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
grep -h "two fp16 pieces max" $o/tests.txt | head -20
for rep in 1 2; do
for dg in 0 1; do
  PIVP_X3_DGRAD=$dg timeout -k 10 200 python bench.py --precision fp16x3 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train_dg$dg.json || exit 1
  python -c "import json; d=json.load(open('$o/train_dg$dg.json')); print('PIVP_X3_DGRAD=$dg train step', d['ms_per_step'])"
done
done
