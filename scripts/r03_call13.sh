#!/bin/bash
# round 3, call 13: op test of the LayerNorm-sums-in-gate-backward kernel, then its per-kernel durations for the three cell shapes
set -o pipefail
mkdir -p gpurun_out/r03/gates_ln
timeout -k 10 300 python -m pytest tests/test_gpu_backward_ops.py -x -q -k layernorm_plus_gate > gpurun_out/r03/gates_ln/test.txt 2>&1 || { tail -30 gpurun_out/r03/gates_ln/test.txt; exit 1; }
tail -2 gpurun_out/r03/gates_ln/test.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for parts in 8; do
for shape in 1024,32 256,64 64,128; do
  d=$R/gpurun_out/r03/gates_ln/p${parts}_${shape/,/x}
  (cd $R && PIVP_LNSUM_PARTS=$parts timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 scripts/bench_gates_ln.py --shape $shape > $d.log 2>&1) || { tail -5 $d.log; exit 1; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== parts $parts shape $shape"
  [ -n "$f" ] || { echo "no kernel_stats.csv under $d"; exit 1; }
  grep -E "lstm_gates_bwd|ln_bwd" "$f" | cut -d, -f1-5 | tee -a $R/gpurun_out/r03/gates_ln/summary.txt
  grep "per call" $d.log | tee -a $R/gpurun_out/r03/gates_ln/summary.txt
  find $d -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $d -name "*.db" -delete
done; done
