#!/bin/bash
# round 4, call 68: fp16x3 mode: lstm5's data gradient (8-wide map) on the ring kernel's fp16 form instead of the fp32 kernel
set -o pipefail
o=gpurun_out/r04/c68
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py tests/test_gpu_train.py tests/test_gpu_configs.py -x -q -s -k "fp16x3 or split or refuse or config" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
grep -h "conv5x5 512->192\|fp16x3 train step\|fp16x3) gradients" $o/tests.txt | cut -c1-250 | head -30
for rep in 1 2 3; do
  timeout -k 10 200 python bench.py --precision fp16x3 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train.json || exit 1
  python -c "import json; d=json.load(open('$o/train.json')); print('fp16x3 train step', d['ms_per_step'])"
done
