#!/bin/bash
# round 4, call 63: fp16x3 mode with its new defaults (weight gradients: two timesteps per launch, half the CUs): tests, train step A/B against the fp32 weight-gradient kernel, config 5
set -o pipefail
o=gpurun_out/r04/c63
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py tests/test_gpu_train.py tests/test_gpu_configs.py -x -q -k "fp16x3 or split or refuse or config" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
for rep in 1 2; do
for wg in 0 1; do
  PIVP_X3_WGRAD=$wg timeout -k 10 200 python bench.py --precision fp16x3 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train_wg$wg.json || exit 1
  python -c "import json; d=json.load(open('$o/train_wg$wg.json')); print('PIVP_X3_WGRAD=$wg train step', d['ms_per_step'])"
done
done
for wg in 0 1; do
  PIVP_X3_WGRAD=$wg timeout -k 10 400 python bench.py --config 5 --precision fp16x3 --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train5_wg$wg.json || exit 1
  python -c "import json; d=json.load(open('$o/train5_wg$wg.json')); print('config 5, PIVP_X3_WGRAD=$wg train step', d['ms_per_step'])"
done
