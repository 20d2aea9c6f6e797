#!/bin/bash
# Round 3, GPU call 1: baseline bench of the round-2 tree, trained weights for the fixtures, stall counters of the bf16 kernels.
set -e -o pipefail
mkdir -p gpurun_out/r03
python3 bench.py --no-cpu-baseline > gpurun_out/r03/bench0.json 2> gpurun_out/r03/bench0.err
cat gpurun_out/r03/bench0.json
python3 tests/golden/train_weights.py --model STP --size 64 --steps 1500 --out gpurun_out/r03/trained_stp64_q8.npz > gpurun_out/r03/train_stp64.log 2>&1
tail -8 gpurun_out/r03/train_stp64.log
python3 tests/golden/train_weights.py --model CDNA --size 128 --steps 500 --freeze model/cdna_kerns/W --out gpurun_out/r03/trained_cdna128_q8.npz > gpurun_out/r03/train_cdna128.log 2>&1
tail -8 gpurun_out/r03/train_cdna128.log
BENCH_ARGS="--precision bf16 --mode train" bash scripts/pmc_stalls.sh bf16train
