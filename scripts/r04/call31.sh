#!/bin/bash
# round 4, call 31: fragment loads in the middle of a k-step's MFMAs (default) against behind them (variant build): layer times of both split modes, tests
set -o pipefail
o=gpurun_out/r04/c31
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -k "fp16x3 or x6" > $o/tests_bf16.txt 2>&1 || { tail -40 $o/tests_bf16.txt; exit 1; }
tail -1 $o/tests_bf16.txt
for m in 6 h3; do
  for lib in physical-interaction-video-prediction_amd/libpivp_hip.so physical-interaction-video-prediction_amd/variants/libpivp_hip_nomid.so; do
    echo "== mode $m $lib" | tee -a $o/layers.txt
    PIVP_BENCH_LIB=$lib PIVP_BENCH_INTERLEAVE=1 PIVP_BENCH_BF16=$m timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 2>&1 | grep -v amdgpu.ids | tee -a $o/layers.txt || exit 1
  done
done
