#!/bin/bash
# round 4, call 58: absmax_partials with 512 threads and eight loads in flight; fp16x3 train step A/B against the three-bf16-piece data gradients
set -o pipefail
o=gpurun_out/r04/c58
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py tests/test_gpu_train.py -x -q -k "fp16x3 or split or refuse" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
for rep in 1 2; do
for dg in 0 1; do
  PIVP_X3_DGRAD=$dg timeout -k 10 200 python bench.py --precision fp16x3 --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $o/train_dg$dg.json || exit 1
  python -c "import json; d=json.load(open('$o/train_dg$dg.json')); print('PIVP_X3_DGRAD=$dg train step', d['ms_per_step'])"
done
done
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o tr -- python3 bench.py --precision fp16x3 --mode train --steps 5 --warmup 3 --no-cpu-baseline --no-roofline > $o/bench.json 2> $o/prof.err || { tail -5 $o/prof.err; exit 1; }
f=$(find $o/prof -name "*kernel_stats.csv" | head -1)
cp $f $o/fp16x3_train_kernel_stats.csv
grep "absmax\|x6g_kernel<4, 2, false" $o/fp16x3_train_kernel_stats.csv | cut -c1-200
rm -rf $o/prof
