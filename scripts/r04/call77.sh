#!/bin/bash
# round 4, call 77: what do the atomics at the end of the 25-tap weight-gradient blocks cost a train step?  (timing-only variant build without them)
set -o pipefail
o=gpurun_out/r04/c77
mkdir -p $o
for rep in 1 2; do
for prec in fp16x3 bf16; do
  timeout -k 10 200 python bench.py --precision $prec --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/t.json || exit 1
  python -c "import json; d=json.load(open('$o/t.json')); print('$prec: default                    train step', d['ms_per_step'])"
  PIVP_BENCH_LIB=physical-interaction-video-prediction_amd/variants/libpivp_hip_noatomic.so timeout -k 10 200 python scripts/r04/bench_with_lib.py --precision $prec --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/t.json || exit 1
  python -c "import json; d=json.load(open('$o/t.json')); print('$prec: no atomics (timing only)   train step', d['ms_per_step'])"
done
done
