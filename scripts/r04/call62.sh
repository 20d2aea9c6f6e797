#!/bin/bash
# round 4, call 62: fp16x3 train step against the timesteps per weight-gradient launch (PIVP_WGRAD_BATCH) and the launch's block target (PIVP_WGB_SLOTS)
set -o pipefail
o=gpurun_out/r04/c62
mkdir -p $o
for gb in 3 2 1; do
for sl in 160 128 96 64; do
  PIVP_WGRAD_BATCH=$gb PIVP_WGB_SLOTS=$sl timeout -k 10 200 python bench.py --precision fp16x3 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train.json || exit 1
  python -c "import json; d=json.load(open('$o/train.json')); print('PIVP_WGRAD_BATCH=$gb PIVP_WGB_SLOTS=$sl train step', d['ms_per_step'])" | tee -a $o/grid.txt
done
done
