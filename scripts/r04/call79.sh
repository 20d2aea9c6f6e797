#!/bin/bash
# round 4, call 79: bf16x6 train step against the timesteps per weight-gradient launch and the launch's block target
set -o pipefail
o=gpurun_out/r04/c79
mkdir -p $o
for gb in 2 3 4 8; do
for sl in 192 128 96; do
  PIVP_WGRAD_BATCH=$gb PIVP_WGB_SLOTS=$sl timeout -k 10 200 python bench.py --precision bf16x6 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train.json || exit 1
  python -c "import json; d=json.load(open('$o/train.json')); print('PIVP_WGRAD_BATCH=$gb PIVP_WGB_SLOTS=$sl train step', d['ms_per_step'])" | tee -a $o/grid.txt
done
done
