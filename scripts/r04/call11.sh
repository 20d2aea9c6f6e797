#!/bin/bash
# round 4, call 11: whole GPU suite with the fused output-side launch in the plan; config 5 geometry and the train step, fused against separate
set -o pipefail
o=gpurun_out/r04/c11
mkdir -p $o
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $o/gpu_tests.txt 2>&1 || { tail -60 $o/gpu_tests.txt; exit 1; }
tail -2 $o/gpu_tests.txt
for fh in 0 1 0 1; do
  PIVP_FRAME_HEAD=$fh timeout -k 10 300 python bench.py --size 128 --seq-len 20 --no-cpu-baseline --no-train --steps 5 --warmup 2 > $o/roll128_fh${fh}.json 2>> $o/err.txt || exit 1
  echo "128x128 T=20 rollout PIVP_FRAME_HEAD=$fh: $(python -c "import json; print(json.loads(open('$o/roll128_fh${fh}.json').read().splitlines()[-1])['ms_per_step'])") ms"
done
for fh in 0 1 0 1; do
  PIVP_FRAME_HEAD=$fh timeout -k 10 200 python bench.py --mode train --no-cpu-baseline --no-roofline --steps 20 > $o/train_fh${fh}.json 2>> $o/err.txt || exit 1
  echo "fp32 train PIVP_FRAME_HEAD=$fh: $(python -c "import json; print(json.loads(open('$o/train_fh${fh}.json').read().splitlines()[-1])['ms_per_step'])") ms"
  PIVP_FRAME_HEAD=$fh timeout -k 10 200 python bench.py --mode train --precision bf16 --no-cpu-baseline --no-roofline --steps 20 > $o/train16_fh${fh}.json 2>> $o/err.txt || exit 1
  echo "bf16 train PIVP_FRAME_HEAD=$fh: $(python -c "import json; print(json.loads(open('$o/train16_fh${fh}.json').read().splitlines()[-1])['ms_per_step'])") ms"
done
for fh in 0 1; do
  PIVP_FRAME_HEAD=$fh timeout -k 10 200 python bench.py --precision bf16 --no-cpu-baseline --no-train --steps 30 > $o/roll16_fh${fh}.json 2>> $o/err.txt || exit 1
  echo "bf16 rollout PIVP_FRAME_HEAD=$fh: $(python -c "import json; print(json.loads(open('$o/roll16_fh${fh}.json').read().splitlines()[-1])['ms_per_step'])") ms"
done
