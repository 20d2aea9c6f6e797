#!/bin/bash
# round 4, call 22: where the three-piece kernel's k-step goes: timing-only variants (scripts/r04/build_x6_variants.sh), lstm1 / lstm4 / lstm7 at B = 32
set -o pipefail
o=gpurun_out/r04/c22
mkdir -p $o
v=physical-interaction-video-prediction_amd/variants
for n in 0 15 7 14 22 24 30; do
  lib=$v/libpivp_hip_x6abl$n.so
  [ $n = 0 ] && lib=physical-interaction-video-prediction_amd/libpivp_hip.so
  echo "== PIVP_X6_ABL=$n" | tee -a $o/abl.txt
  PIVP_BENCH_LIB=$lib PIVP_BENCH_BF16=6 timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 lstm1,lstm4,lstm7 2>&1 | grep -v amdgpu.ids | tee -a $o/abl.txt || exit 1
done
