#!/bin/bash
# kernel trace of the fp32 train step on ONE stream (PIVP_SIDE_STREAM=0: durations are additive), grouped by kernel and grid:
# bash scripts/prof_train_single.sh <tag> [extra bench flags]
set -e
tag=${1:-cur}; shift || true
out=gpurun_out/prof_single_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PIVP_SIDE_STREAM=0
rocprofv3 --kernel-trace --output-format csv -d $out/kt -o train -- python3 bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $out/log 2>&1
python3 scripts/trace_by_grid.py $out/kt/train_kernel_trace.csv > $out/by_grid.txt
rm -f $out/kt/*kernel_trace.csv
cat $out/by_grid.txt
