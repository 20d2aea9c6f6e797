#!/bin/bash
# kernel trace of the fp32 train step: bash scripts/prof_train.sh <tag> [extra bench flags]
set -e
tag=${1:-cur}; shift || true
out=gpurun_out/prof_train_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o train -- python3 bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $out/log 2>&1
python3 scripts/overlap_report.py $out/kt/train_kernel_trace.csv > $out/overlap.txt 2>&1 || true
python3 scripts/trace_by_grid.py $out/kt/train_kernel_trace.csv > $out/by_grid.txt 2>&1 || true
rm -f $out/kt/*kernel_trace.csv
cat $out/overlap.txt
head -45 $out/kt/train_kernel_stats.csv | cut -c1-160
