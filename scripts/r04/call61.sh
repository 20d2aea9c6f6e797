#!/bin/bash
# round 4, call 61: fp16x3 train step against the block target of the fp16-piece weight gradient (PIVP_WGB_SLOTS; default = CUs)
set -o pipefail
o=gpurun_out/r04/c61
mkdir -p $o
for rep in 1 2; do
for sl in 256 192 128 96 64; do
  PIVP_WGB_SLOTS=$sl timeout -k 10 200 python bench.py --precision fp16x3 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train_sl$sl.json || exit 1
  python -c "import json; d=json.load(open('$o/train_sl$sl.json')); print('PIVP_WGB_SLOTS=$sl train step', d['ms_per_step'])"
done
done
