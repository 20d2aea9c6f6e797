#!/bin/bash
# round 4, call 81: the 25-tap weight gradient with two sets of LDS images (one barrier per tile, the next tile staged between the k-steps) against one set (variant build)
set -o pipefail
o=gpurun_out/r04/c81
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -k "wgrad" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
for rep in 1 2; do
for prec in fp16x3 bf16; do
  timeout -k 10 200 python bench.py --precision $prec --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/t.json || exit 1
  python -c "import json; d=json.load(open('$o/t.json')); print('$prec: two image sets (default)  train step', d['ms_per_step'])"
  PIVP_BENCH_LIB=physical-interaction-video-prediction_amd/variants/libpivp_hip_nodbuf.so timeout -k 10 200 python scripts/r04/bench_with_lib.py --precision $prec --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/t.json || exit 1
  python -c "import json; d=json.load(open('$o/t.json')); print('$prec: one image set (variant)   train step', d['ms_per_step'])"
done
done
