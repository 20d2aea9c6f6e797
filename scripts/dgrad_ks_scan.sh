#!/bin/bash
# ConvLSTM data gradients under every forced K split (PIVP_DGRAD_KS = 1..8; the tile is still the cost model's for that split):
# single-stream kernel trace of the train step per setting, the non-LSTM igemm_f32 launches grouped by grid.
# bash scripts/dgrad_ks_scan.sh  ->  gpurun_out/dgrad_scan/ks<k>.txt (+ shapes<k>.txt: which tile / split each shape got)
set -e
out=gpurun_out/dgrad_scan
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PIVP_SIDE_STREAM=0
for k in 0 1 2 3 4 5 6 8; do
  if [ $k != 0 ]; then export PIVP_DGRAD_KS=$k; fi
  PIVP_DGRAD_DEBUG=1 python3 bench.py --mode train --steps 1 --warmup 0 --no-cpu-baseline --no-roofline 2>&1 >/dev/null | grep "^dgrad" | sort | uniq -c > $out/shapes$k.txt
  rocprofv3 --kernel-trace --output-format csv -d $out/kt$k -o t -- python3 bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $out/log$k 2>&1
  python3 scripts/trace_by_grid.py $out/kt$k/t_kernel_trace.csv | grep -E "false|total" > $out/ks$k.txt
  rm -rf $out/kt$k
  echo "== ks $k"; cat $out/shapes$k.txt; cat $out/ks$k.txt
done
