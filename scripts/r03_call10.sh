#!/bin/bash
# Round 3, GPU call 10: do write-through (sc1) stores of c / h in the ConvLSTM epilogue shorten the kernel boundary behind it?
set -e -o pipefail
mkdir -p gpurun_out/r03
for aux in 0 16 0 16; do
  PIVP_EXTRA_FLAGS="-DPIVP_ST_AUX=$aux" python3 physical-interaction-video-prediction_amd/build.py --force > /dev/null 2>&1
  python3 bench.py --no-cpu-baseline --no-train > gpurun_out/r03/bench10_aux$aux.json 2> /dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r03/bench10_aux$aux.json'));print('aux $aux: rollout', d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg launch us', d['roofline']['avg_launch_us'])"
done
