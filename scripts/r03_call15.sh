#!/bin/bash
# round 3, call 15: s_setprio 3 in the main stream's kernels (-DPIVP_MAIN_PRIO=3), now that the bf16 sweep's critical path is the main
# stream alone (round 2 measured it when the bf16 weight gradients were the long pole): train step fp32 / bf16 and the rollout, A/B/A/B
set -e -o pipefail
o=gpurun_out/r03/main_prio
mkdir -p $o
for prio in 0 3 0 3; do
  PIVP_EXTRA_FLAGS="-DPIVP_MAIN_PRIO=$prio" python3 physical-interaction-video-prediction_amd/build.py --force > $o/build.log 2>&1
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench_prio$prio.json 2> $o/err.txt
  python3 - <<PY
import json
d=json.loads(open('$o/bench_prio$prio.json').read().strip().splitlines()[-1])
print('PIVP_MAIN_PRIO=$prio: rollout %.3f ms, train %.3f ms, train_bf16 %.3f ms' % (d['ms_per_step'], d['train']['ms_per_step'], d['train_bf16']['ms_per_step']), flush=True)
PY
done
PIVP_EXTRA_FLAGS="" python3 physical-interaction-video-prediction_amd/build.py --force > $o/build.log 2>&1
