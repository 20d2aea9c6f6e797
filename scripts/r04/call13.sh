#!/bin/bash
# round 4, call 13: fp32 ConvLSTM weight gradient with THREE resident blocks per CU (PIVP_WGRAD_OCC=3): gradient tests, train step A/B, kernel stats
set -o pipefail
o=gpurun_out/r04/c13
mkdir -p $o
PIVP_WGRAD_OCC=3 timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward_ops.py -m gpu -x -q > $o/tests.txt 2>&1 || { tail -80 $o/tests.txt; exit 1; }
tail -2 $o/tests.txt
for rep in 1 2; do
  for occ in 2 3; do
    PIVP_WGRAD_OCC=$occ timeout -k 10 200 python bench.py --mode train --no-cpu-baseline --no-roofline --steps 20 > $o/train_occ${occ}_$rep.json 2>> $o/err.txt || exit 1
    echo "fp32 train PIVP_WGRAD_OCC=$occ rep $rep: $(python -c "import json; print(json.loads(open('$o/train_occ${occ}_$rep.json').read().splitlines()[-1])['ms_per_step'])") ms"
  done
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for occ in 2 3; do
  PIVP_WGRAD_OCC=$occ PIVP_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt$occ -o r -- python3 bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $o/kt$occ.log 2>&1 || exit 1
  rm -f $o/kt$occ/*kernel_trace.csv
  echo "== single stream, PIVP_WGRAD_OCC=$occ"; grep "wgrad5x5" $o/kt$occ/*kernel_stats.csv | cut -c1-160
done
