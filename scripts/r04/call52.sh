#!/bin/bash
# round 4, call 52: is the "cost of the B fragment loads" (ablation 8) really the loads, or the operand values?  The default fp16x3 kernels on random / constant / zero operands
set -o pipefail
o=gpurun_out/r04/c52
mkdir -p $o
for data in random const zero; do
  echo "== data=$data" | tee -a $o/layers.txt
  PIVP_BENCH_DATA=$data PIVP_X3_TH16=0 PIVP_BENCH_INTERLEAVE=1 PIVP_BENCH_BF16=h3 timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 2>&1 | grep -v amdgpu.ids | tee -a $o/layers.txt || exit 1
done
for data in random zero; do
  echo "== three pieces, data=$data" | tee -a $o/layers.txt
  PIVP_BENCH_DATA=$data PIVP_BENCH_INTERLEAVE=1 PIVP_BENCH_BF16=6 timeout -k 10 120 python scripts/bench_lstm_layers.py 32 20 2>&1 | grep -v amdgpu.ids | tee -a $o/layers.txt || exit 1
done
