#!/bin/bash
# round 4, call 14: bf16 gate conv with the deepest weight ring that fits (6-7 taps of LDS-DMA in flight) against round 3's 3 taps
set -o pipefail
o=gpurun_out/r04/c14
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q > $o/tests.txt 2>&1 || { tail -80 $o/tests.txt; exit 1; }
tail -2 $o/tests.txt
run() {   # $1 = label
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --precision bf16 --no-cpu-baseline --no-train --steps 30 > $o/roll_$1_$rep.json 2>> $o/err.txt || exit 1
    python - <<PY
import json
d=json.loads(open('$o/roll_$1_$rep.json').read().splitlines()[-1])
print('bf16 rollout [$1] rep $rep: %.3f ms  conv frac %.4f  %s' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_layer_tflops']))
PY
    timeout -k 10 200 python bench.py --mode train --precision bf16 --no-cpu-baseline --no-roofline --steps 20 > $o/train_$1_$rep.json 2>> $o/err.txt || exit 1
    echo "bf16 train [$1] rep $rep: $(python -c "import json; print(json.loads(open('$o/train_$1_$rep.json').read().splitlines()[-1])['ms_per_step'])") ms"
  done
}
run deep
PIVP_BENCH_BF16=1 timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 2>&1 | grep -v amdgpu.ids | tee $o/layers_deep.txt
PIVP_BENCH_BF16=1 timeout -k 10 120 python scripts/bench_lstm_layers.py 256 10 2>&1 | grep -v amdgpu.ids | tee $o/layers_deep_b256.txt
PIVP_EXTRA_FLAGS=-DPIVP_BF16_DEPTH=3 timeout -k 10 600 python physical-interaction-video-prediction_amd/build.py > $o/build3.log 2>&1 || { tail -20 $o/build3.log; exit 1; }
run depth3
PIVP_BENCH_BF16=1 timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 2>&1 | grep -v amdgpu.ids | tee $o/layers_depth3.txt
PIVP_BENCH_BF16=1 timeout -k 10 120 python scripts/bench_lstm_layers.py 256 10 2>&1 | grep -v amdgpu.ids | tee $o/layers_depth3_b256.txt
