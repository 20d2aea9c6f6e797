#!/bin/bash
# Shader clock and socket power (rocm-smi) while the rollout loop / the train-step loop run: is the fp32 path's distance from the NOMINAL
# 157.3 TFLOP/s (2.4 GHz) partly the power cap?  (scripts/wgrad_data_dependence.py: the backward kernels run 14 % faster on zero operands.)
set -o pipefail
o=gpurun_out/r03/clocks
mkdir -p $o
sample() {   # $1 = tag, samples while the background job $2 lives (at most 12 s)
  for i in $(seq 1 24); do
    kill -0 $2 2>/dev/null || break
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' ' >> $o/$1.txt; echo >> $o/$1.txt
    sleep 0.5
  done
}
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" > $o/idle.txt
python3 bench.py --steps 1200 --warmup 5 --no-cpu-baseline --no-roofline --no-train > $o/rollout.json 2> $o/rollout.err &
p=$!; sleep 6; sample rollout $p; wait $p
python3 bench.py --mode train --steps 350 --warmup 5 --no-cpu-baseline --no-roofline > $o/train.json 2> $o/train.err &
p=$!; sleep 6; sample train $p; wait $p
python3 bench.py --mode train --precision bf16 --steps 800 --warmup 5 --no-cpu-baseline --no-roofline > $o/train_bf16.json 2> $o/train_bf16.err &
p=$!; sleep 6; sample train_bf16 $p; wait $p
for t in idle rollout train train_bf16; do echo "== $t"; tail -4 $o/$t.txt | cut -c1-300; done
