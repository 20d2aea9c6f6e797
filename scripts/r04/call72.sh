#!/bin/bash
# round 4, call 72: ln_bwd_sums_params_kernel with two samples in flight per trip (64 registers, no spills: a wave fits beside the weight-gradient blocks) against four (88)
set -o pipefail
o=gpurun_out/r04/c72
mkdir -p $o
for rep in 1 2; do
for prec in bf16 fp32 fp16x3; do
  timeout -k 10 200 python bench.py --precision $prec --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/t.json || exit 1
  python -c "import json; d=json.load(open('$o/t.json')); print('$prec: four samples in flight (default)  train step', d['ms_per_step'])"
  PIVP_BENCH_LIB=physical-interaction-video-prediction_amd/variants/libpivp_hip_lnbnj2.so timeout -k 10 200 python scripts/r04/bench_with_lib.py --precision $prec --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/t.json || exit 1
  python -c "import json; d=json.load(open('$o/t.json')); print('$prec: two samples in flight (variant)   train step', d['ms_per_step'])"
done
done
