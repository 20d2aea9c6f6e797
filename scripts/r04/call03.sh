#!/bin/bash
# round 4, call 3: the square 64 x 64 ConvLSTM tile (variants 5 / 6): parity tests, per-layer times against the current choice, rollout A/B
set -o pipefail
o=gpurun_out/r04/c03
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "convlstm" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -2 $o/tests.txt
for v in 0 5 6 3 2 4; do
  echo "== PIVP_LSTM_VARIANT=$v (interleaved, as in a rollout)"
  PIVP_BENCH_INTERLEAVE=1 PIVP_LSTM_VARIANT=$v timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 lstm3,lstm4,lstm5,lstm6 2>&1 | grep -v amdgpu.ids | tee -a $o/layers_v$v.txt
done
for rep in 1 2; do
  for sq in 0 1 2; do
    PIVP_LSTM_SQ=$sq timeout -k 10 200 python bench.py --no-cpu-baseline --no-train --steps 30 > $o/roll_sq${sq}_$rep.json 2>> $o/err.txt || exit 1
    python - <<PY
import json
d=json.loads(open('$o/roll_sq${sq}_$rep.json').read().splitlines()[-1])
print('rollout PIVP_LSTM_SQ=$sq rep $rep: %.3f ms  frac %.4f  %s' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_layer_tflops']))
PY
  done
done
