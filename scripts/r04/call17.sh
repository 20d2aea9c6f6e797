#!/bin/bash
# round 4, call 17: frame_head with the 1x1 mixes on the fp32 MFMA: bit-identity tests, rollout A/B, kernel statistics, phase stamps
set -o pipefail
o=gpurun_out/r04/c17
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "frame_head" > $o/tests_ops.txt 2>&1 || { tail -60 $o/tests_ops.txt; exit 1; }
tail -2 $o/tests_ops.txt
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_trained.py -m gpu -x -q > $o/tests_model.txt 2>&1 || { tail -60 $o/tests_model.txt; exit 1; }
tail -2 $o/tests_model.txt
for rep in 1 2; do
  for fh in 0 1; do
    PIVP_FRAME_HEAD=$fh timeout -k 10 200 python bench.py --no-cpu-baseline --no-train --steps 30 > $o/roll_fh${fh}_$rep.json 2>> $o/err.txt || exit 1
    echo "rollout PIVP_FRAME_HEAD=$fh rep $rep: $(python -c "import json; print(json.loads(open('$o/roll_fh${fh}_$rep.json').read().splitlines()[-1])['ms_per_step'])") ms"
  done
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PIVP_FRAME_HEAD=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt1 -o r -- python3 bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-train --no-roofline > $o/kt1.log 2>&1 || exit 1
rm -f $o/kt1/*kernel_trace.csv
grep "frame_head\|skinny" $o/kt1/*kernel_stats.csv | cut -c1-200
PIVP_EXTRA_FLAGS=-DPIVP_FH_STAMPS timeout -k 10 600 python physical-interaction-video-prediction_amd/build.py > $o/build.log 2>&1 || { tail -20 $o/build.log; exit 1; }
timeout -k 10 300 python scripts/r04/fh_stamps.py 2>&1 | grep -v amdgpu.ids | tee $o/fh_stamps.txt
