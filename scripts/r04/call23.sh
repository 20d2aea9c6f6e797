#!/bin/bash
# round 4, call 23: block-level phase stamps of the three-piece kernel (stamped variant build)
set -o pipefail
o=gpurun_out/r04/c23
mkdir -p $o
PIVP_STAMP_MODE=6 PIVP_BENCH_LIB=physical-interaction-video-prediction_amd/variants/libpivp_hip_x6ablstamps.so timeout -k 10 120 python scripts/bf16_stamps.py 2>&1 | grep -v amdgpu.ids | tee $o/x6_stamps.txt
PIVP_STAMP_MODE=1 PIVP_BENCH_LIB=physical-interaction-video-prediction_amd/variants/libpivp_hip_x6ablstamps.so timeout -k 10 120 python scripts/bf16_stamps.py 2>&1 | grep -v amdgpu.ids | tee $o/bf16_stamps.txt
