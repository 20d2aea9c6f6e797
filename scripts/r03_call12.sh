#!/bin/bash
# round 3, call 12: hipGraph replay of the rollout / train step against direct enqueueing (one case per process)
set -o pipefail
mkdir -p gpurun_out/r03
o=gpurun_out/r03/graph_replay.txt
: > $o
timeout -k 10 200 python scripts/graph_replay.py --precision bf16 --leg rollout >> $o 2>&1; echo "exit $?" >> $o
echo "--- single stream (PIVP_SIDE_STREAM=0)" >> $o
PIVP_SIDE_STREAM=0 timeout -k 10 200 python scripts/graph_replay.py --precision bf16 --leg train >> $o 2>&1; echo "exit $?" >> $o
PIVP_SIDE_STREAM=0 timeout -k 10 200 python scripts/graph_replay.py --precision fp32 --leg train >> $o 2>&1; echo "exit $?" >> $o
grep -v amdgpu.ids $o | cut -c1-400
