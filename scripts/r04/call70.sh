#!/bin/bash
# round 4, call 70: fp16x3 train step with an odd batch (lstm5 on the fp32 kernels), the mode's other train tests
set -o pipefail
o=gpurun_out/r04/c70
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train.py -x -q -s -k "fp16x3 or split" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
grep -h "fp16x3 train step" $o/tests.txt
