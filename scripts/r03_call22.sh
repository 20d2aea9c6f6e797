#!/bin/bash
# round 3, call 22: the fp32 ConvLSTM forward kernel on random / constant / zero operands (does the sustained MFMA rate depend on the data?),
# back to back per layer; also the bf16 kernel
set -o pipefail
o=gpurun_out/r03/data_dependence
mkdir -p $o
for d in random const zero random; do
  PIVP_BENCH_DATA=$d timeout -k 10 120 python scripts/bench_lstm_layers.py 32 30 2>&1 | grep -v amdgpu.ids > $o/f32_$d.txt || { tail $o/f32_$d.txt; exit 1; }
  echo "== fp32, data $d"; tail -9 $o/f32_$d.txt
done
for d in random zero; do
  PIVP_BENCH_BF16=1 PIVP_BENCH_DATA=$d timeout -k 10 120 python scripts/bench_lstm_layers.py 32 30 2>&1 | grep -v amdgpu.ids > $o/bf16_$d.txt || { tail $o/bf16_$d.txt; exit 1; }
  echo "== bf16, data $d"; tail -9 $o/bf16_$d.txt
done
