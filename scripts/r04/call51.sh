#!/bin/bash
# round 4, call 51: fp16x3 gate kernels on tiles of 16 x 16 anchors (256 anchors x 16 channels per block: half the weight bytes out of L2 per multiply-add)
set -o pipefail
o=gpurun_out/r04/c51
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py -x -q -k "fp16x3 or split" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
for th in 0 1; do
  echo "== PIVP_X3_TH16=$th" | tee -a $o/layers.txt
  PIVP_X3_TH16=$th PIVP_BENCH_INTERLEAVE=1 PIVP_BENCH_BF16=h3 timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 2>&1 | grep -v amdgpu.ids | tee -a $o/layers.txt || exit 1
done
for th in 0 1; do
  PIVP_X3_TH16=$th timeout -k 10 200 python bench.py --precision fp16x3 --steps 30 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | tail -1 > $o/bench_th$th.json || exit 1
  python -c "import json; d=json.load(open('$o/bench_th$th.json')); print('TH16=$th', d['ms_per_step'], d['value'])"
done
