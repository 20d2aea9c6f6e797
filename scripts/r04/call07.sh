#!/bin/bash
# round 4, call 7: where frame_head_kernel's 60 us go: a stamped build (PIVP_FH_STAMPS), one block's phases per wave
set -o pipefail
o=gpurun_out/r04/c07
mkdir -p $o
PIVP_EXTRA_FLAGS=-DPIVP_FH_STAMPS timeout -k 10 600 python physical-interaction-video-prediction_amd/build.py > $o/build.log 2>&1 || { tail -20 $o/build.log; exit 1; }
timeout -k 10 300 python scripts/r04/fh_stamps.py 2>&1 | grep -v amdgpu.ids | tee $o/fh_stamps.txt
