#!/bin/bash
# round 4, call 47: timing-only variants of the L2-direct fp16 kernels: no B loads (8), no A lo-plane reads (4), neither (12)
set -o pipefail
o=gpurun_out/r04/c49
mkdir -p $o
for v in "" abl32 abl8; do
  lib=physical-interaction-video-prediction_amd/libpivp_hip.so
  [ -n "$v" ] && lib=physical-interaction-video-prediction_amd/variants/libpivp_hip_$v.so
  echo "== ${v:-default}" | tee -a $o/layers.txt
  PIVP_BENCH_LIB=$lib PIVP_BENCH_INTERLEAVE=1 PIVP_BENCH_BF16=h3 timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 lstm1,lstm4,lstm7 2>&1 | grep -v amdgpu.ids | tee -a $o/layers.txt || exit 1
done
