#!/bin/bash
# round 4, call 2: fp32 train step against the ConvLSTM weight gradients' timestep batch (PIVP_WGRAD_BATCH; the existing kernel-row kernel
# already takes tcount > 1), interleaved on one box; then the two-queue breakdown of the default.
set -o pipefail
o=gpurun_out/r04/c02
mkdir -p $o
T="--mode train --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
for rep in 1 2; do
  for gb in 1 2 4 8; do
    PIVP_WGRAD_BATCH=$gb timeout -k 10 200 python bench.py $T > $o/train_gb${gb}_$rep.json 2>> $o/err.txt || exit 1
    echo "fp32 train  PIVP_WGRAD_BATCH=$gb rep $rep: $(python -c "import json,sys; print(json.loads(open('$o/train_gb${gb}_$rep.json').read().splitlines()[-1])['ms_per_step'])") ms"
  done
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $o/train -o train -- python3 bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $o/train.log 2>&1 || exit 1
python3 scripts/queue_breakdown.py $o/train/train_kernel_trace.csv 3 0.45 > $o/train_queues.txt 2>&1
python3 scripts/overlap_report.py $o/train/train_kernel_trace.csv 0.45 > $o/train_overlap.txt 2>&1
rm -f $o/train/*kernel_trace.csv
cat $o/train_queues.txt | head -60
