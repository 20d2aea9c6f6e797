#!/bin/bash
# Round 3, GPU call 7: bf16 gate conv with fragments requested two k-steps ahead (three register sets): parity, rollout at B = 32 / 256, train step, per-block stamps.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python3 -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/r03/pytest7.log 2>&1 || echo "TESTS FAILED (see gpurun_out/r03/pytest7.log)"
tail -5 gpurun_out/r03/pytest7.log
set -e
for b in 32 256; do
python3 bench.py --precision bf16 --no-train --no-cpu-baseline --batch $b --steps 10 --warmup 3 > gpurun_out/r03/bf16_7_b$b.json 2> gpurun_out/r03/bf16_7_b$b.err
python3 -c "import json;d=json.load(open('gpurun_out/r03/bf16_7_b$b.json'));print('bf16 rollout B=$b', d['ms_per_step'], 'ms', d['roofline']['frac'], d['roofline']['per_layer_tflops'])"
done
python3 bench.py --precision bf16 --mode train --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > gpurun_out/r03/bf16train7.json 2> gpurun_out/r03/bf16train7.err
python3 -c "import json;d=json.load(open('gpurun_out/r03/bf16train7.json'));print('bf16 train', d['ms_per_step'])"
PIVP_EXTRA_FLAGS="-DPIVP_BF16_STAMPS" python3 physical-interaction-video-prediction_amd/build.py --force > gpurun_out/r03/build_stamps7.log 2>&1
python3 scripts/bf16_stamps.py > gpurun_out/r03/bf16_stamps_blocks7.txt 2>&1
grep -v amdgpu gpurun_out/r03/bf16_stamps_blocks7.txt | grep "launch\|tap loop"
