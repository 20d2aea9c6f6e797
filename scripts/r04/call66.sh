#!/bin/bash
# round 4, call 66: bf16 mode (config 3) train step against the timesteps per weight-gradient launch and the 25-tap kernel's block target
set -o pipefail
o=gpurun_out/r04/c66
mkdir -p $o
echo "default:" | tee -a $o/grid.txt
timeout -k 10 200 python bench.py --precision bf16 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train.json || exit 1
python -c "import json; d=json.load(open('$o/train.json')); print('defaults train step', d['ms_per_step'])" | tee -a $o/grid.txt
for gb in 8 4 2; do
for sl in 256 192 128 96; do
  PIVP_WGRAD_BATCH=$gb PIVP_WGB_SLOTS=$sl timeout -k 10 200 python bench.py --precision bf16 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train.json || exit 1
  python -c "import json; d=json.load(open('$o/train.json')); print('PIVP_WGRAD_BATCH=$gb PIVP_WGB_SLOTS=$sl train step', d['ms_per_step'])" | tee -a $o/grid.txt
done
done
