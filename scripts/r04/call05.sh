#!/bin/bash
# round 4, call 5: kernel-trace statistics of the rollout with and without the fused output-side launch
set -o pipefail
o=gpurun_out/r04/c05
mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for fh in 1 0; do
  export PIVP_FRAME_HEAD=$fh
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt$fh -o r -- python3 bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-train --no-roofline > $o/kt$fh.log 2>&1 || exit 1
  rm -f $o/kt$fh/*kernel_trace.csv
  echo "== PIVP_FRAME_HEAD=$fh"; head -14 $o/kt$fh/*kernel_stats.csv | cut -c1-160
done
