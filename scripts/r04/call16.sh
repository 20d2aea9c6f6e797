#!/bin/bash
# round 4, call 16 (the Linear blocks first in the grid): enc4 and the motion head's partial sums in one grid (PIVP_ENC4_PARTIALS=1 / 0): model tests, rollout A/B, kernel statistics
set -o pipefail
o=gpurun_out/r04/c16
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q > $o/tests.txt 2>&1 || { tail -60 $o/tests.txt; exit 1; }
tail -2 $o/tests.txt
for rep in 1 2; do
  for f in 0 1; do
    PIVP_ENC4_PARTIALS=$f timeout -k 10 200 python bench.py --no-cpu-baseline --no-train --steps 30 > $o/roll_f${f}_$rep.json 2>> $o/err.txt || exit 1
    echo "rollout PIVP_ENC4_PARTIALS=$f rep $rep: $(python -c "import json; print(json.loads(open('$o/roll_f${f}_$rep.json').read().splitlines()[-1])['ms_per_step'])") ms"
  done
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt1 -o r -- python3 bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-train --no-roofline > $o/kt1.log 2>&1 || exit 1
rm -f $o/kt1/*kernel_trace.csv
grep -v "igemm_f32_kernel" $o/kt1/*kernel_stats.csv | head -14 | cut -c1-180
