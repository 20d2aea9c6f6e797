#!/bin/bash
# round 4, call 71: ln_bwd_sums_params_kernel held to 72 registers (a wave fits beside the bf16 mode's weight-gradient blocks) against 88 (variant build)
set -o pipefail
o=gpurun_out/r04/c71
mkdir -p $o
for rep in 1 2; do
for prec in bf16 fp32 fp16x3; do
  timeout -k 10 200 python bench.py --precision $prec --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/t.json || exit 1
  python -c "import json; d=json.load(open('$o/t.json')); print('$prec: 72 registers (default build)  train step', d['ms_per_step'])"
  PIVP_BENCH_LIB=physical-interaction-video-prediction_amd/variants/libpivp_hip_lnb1.so timeout -k 10 200 python scripts/r04/bench_with_lib.py --precision $prec --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/t.json || exit 1
  python -c "import json; d=json.load(open('$o/t.json')); print('$prec: 88 registers (variant)        train step', d['ms_per_step'])"
done
done
