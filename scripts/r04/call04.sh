#!/bin/bash
# round 4, call 4: the fused output-side launch (frame_head): op-level bit-identity test, the model-level suites, rollout / train A/B
set -o pipefail
o=gpurun_out/r04/c04
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "frame_head or heads or composite or cdna or stp" > $o/tests_ops.txt 2>&1 || { tail -60 $o/tests_ops.txt; exit 1; }
tail -2 $o/tests_ops.txt
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_trained.py tests/test_gpu_configs.py tests/test_gpu_edge_cases.py -m gpu -x -q > $o/tests_model.txt 2>&1 || { tail -60 $o/tests_model.txt; exit 1; }
tail -2 $o/tests_model.txt
for rep in 1 2; do
  for fh in 0 1; do
    PIVP_FRAME_HEAD=$fh timeout -k 10 200 python bench.py --no-cpu-baseline --no-train --steps 30 > $o/roll_fh${fh}_$rep.json 2>> $o/err.txt || exit 1
    echo "rollout PIVP_FRAME_HEAD=$fh rep $rep: $(python -c "import json; print(json.loads(open('$o/roll_fh${fh}_$rep.json').read().splitlines()[-1])['ms_per_step'])") ms"
  done
done
for fh in 0 1; do
  PIVP_FRAME_HEAD=$fh timeout -k 10 200 python bench.py --model STP --no-cpu-baseline --no-train --steps 30 > $o/roll_stp_fh${fh}.json 2>> $o/err.txt || exit 1
  echo "STP rollout PIVP_FRAME_HEAD=$fh: $(python -c "import json; print(json.loads(open('$o/roll_stp_fh${fh}.json').read().splitlines()[-1])['ms_per_step'])") ms"
  PIVP_FRAME_HEAD=$fh timeout -k 10 200 python bench.py --precision bf16 --no-cpu-baseline --no-train --steps 30 > $o/roll_bf16_fh${fh}.json 2>> $o/err.txt || exit 1
  echo "bf16 rollout PIVP_FRAME_HEAD=$fh: $(python -c "import json; print(json.loads(open('$o/roll_bf16_fh${fh}.json').read().splitlines()[-1])['ms_per_step'])") ms"
  PIVP_FRAME_HEAD=$fh timeout -k 10 200 python bench.py --mode train --no-cpu-baseline --no-roofline --steps 20 > $o/train_fh${fh}.json 2>> $o/err.txt || exit 1
  echo "fp32 train PIVP_FRAME_HEAD=$fh: $(python -c "import json; print(json.loads(open('$o/train_fh${fh}.json').read().splitlines()[-1])['ms_per_step'])") ms"
  PIVP_FRAME_HEAD=$fh timeout -k 10 300 python bench.py --size 128 --seq-len 20 --no-cpu-baseline --no-train --steps 5 --warmup 2 > $o/roll128_fh${fh}.json 2>> $o/err.txt || exit 1
  echo "128x128 T=20 rollout PIVP_FRAME_HEAD=$fh: $(python -c "import json; print(json.loads(open('$o/roll128_fh${fh}.json').read().splitlines()[-1])['ms_per_step'])") ms"
done
