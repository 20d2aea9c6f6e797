#!/bin/bash
# round 4, call 29: wave priority 3 for the main stream's kernels (-DPIVP_MAIN_PRIO=3, whole-library variant) in the three train steps: since the
# three-piece mode the main stream is the critical path and the side stream has 10 ms of slack per step (round 2 measured no gain for the fp32 kernels)
set -o pipefail
o=gpurun_out/r04/c29
mkdir -p $o
v=physical-interaction-video-prediction_amd/variants/libpivp_hip_prio3.so
for p in bf16x6 fp32 bf16; do
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --precision $p --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $o/base_${p}_$rep.json 2> $o/err.txt || { tail -5 $o/err.txt; exit 1; }
    PIVP_BENCH_LIB=$v timeout -k 10 200 python scripts/r04/bench_with_lib.py --precision $p --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $o/prio3_${p}_$rep.json 2> $o/err.txt || { tail -5 $o/err.txt; exit 1; }
  done
done
python - <<'EOF2'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04/c29/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['ms_per_step'])
EOF2
