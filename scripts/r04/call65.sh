#!/bin/bash
# round 4, call 65: kernel trace of the fp16x3 train step with fp16-piece data and weight gradients
set -o pipefail
o=gpurun_out/r04/c65
mkdir -p $o
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o tr -- python3 bench.py --precision fp16x3 --mode train --steps 5 --warmup 3 --no-cpu-baseline --no-roofline > $o/bench.json 2> $o/prof.err || { tail -5 $o/prof.err; exit 1; }
f=$(find $o/prof -name "*kernel_stats.csv" | head -1)
cp $f $o/fp16x3_train_kernel_stats.csv
rm -rf $o/prof
head -5 $o/fp16x3_train_kernel_stats.csv | cut -c1-160
