#!/bin/bash
# round 4, call 34: kernel statistics of the fp32 rollout alone (call 33's trace also held the split modes' rollout legs) and the fp32 counters
set -o pipefail
out=gpurun_out/prof_v3b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NB="--no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o rollout -- python3 bench.py --steps 7 --warmup 2 $NB --no-train --no-bf16x6 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
rm -f $out/kt/*kernel_trace.csv
head -5 $out/kt/rollout_kernel_stats.csv | cut -c1-120
