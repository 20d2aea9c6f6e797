#!/bin/bash
# round 4, call 50 (sharing per TAP: one barrier per tap; was call 48): fp16 kernels: every B fragment loaded once per block and shared through LDS (default build) against one copy per wave from L2 (variant)
set -o pipefail
o=gpurun_out/r04/c50
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py -x -q -k "fp16x3" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
for lib in physical-interaction-video-prediction_amd/libpivp_hip.so physical-interaction-video-prediction_amd/variants/libpivp_hip_noshare.so; do
  echo "== $lib" | tee -a $o/layers.txt
  PIVP_BENCH_LIB=$lib PIVP_BENCH_INTERLEAVE=1 PIVP_BENCH_BF16=h3 timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 2>&1 | grep -v amdgpu.ids | tee -a $o/layers.txt || exit 1
done
timeout -k 10 120 python scripts/r04/soak_split.py 60 2>&1 | grep -v amdgpu.ids | tail -6
