#!/bin/bash
# round 3, call 24: composite_bwd_stp with one block per sample and the whole frame as its LDS window (PIVP_STP_WHOLE=1) against one block per tile (0)
set -o pipefail
o=gpurun_out/r03/stp_whole
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward_ops.py tests/test_gpu_trained.py -x -q -k "stp or STP or composite" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -2 $o/tests.txt
for v in 0 1 2 4 8 4; do
  PIVP_STP_WHOLE=$v timeout -k 10 300 python bench.py --model STP --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $o/train_$v.json 2>$o/err.txt || { tail $o/err.txt; exit 1; }
  python - <<PY
import json
a=json.loads(open('$o/train_$v.json').read().strip().splitlines()[-1])
print('PIVP_STP_WHOLE=$v: STP train %.3f ms' % a['ms_per_step'], flush=True)
PY
done
