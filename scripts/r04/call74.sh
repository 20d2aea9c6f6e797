#!/bin/bash
# round 4, call 74: fp16x3 train step on ONE stream (every weight gradient behind its data gradient) against the side stream
set -o pipefail
o=gpurun_out/r04/c74
mkdir -p $o
run() { timeout -k 10 200 python bench.py --precision fp16x3 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/t.json || exit 1; python -c "import json; d=json.load(open('$o/t.json')); print('$1 train step', d['ms_per_step'])"; }
for rep in 1 2; do
run "default (side stream, 2 timesteps per launch, 128 blocks)"
PIVP_SIDE_STREAM=0 run "one stream, 2 per launch, 128 blocks"
PIVP_SIDE_STREAM=0 PIVP_WGB_SLOTS=256 run "one stream, 2 per launch, 256 blocks"
PIVP_SIDE_STREAM=0 PIVP_WGB_SLOTS=256 PIVP_WGRAD_BATCH=8 run "one stream, 8 per launch, 256 blocks"
done
