#!/bin/bash
# Round 3, GPU call 4: the 25-tap batched bf16 weight gradient (parity, per-layer rate, train step A/B), trained-weight fixtures, bf16 conv at large batch.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest4.log 2>&1 || echo "TESTS FAILED (see gpurun_out/r03/pytest4.log)"
tail -4 gpurun_out/r03/pytest4.log
grep -h "per-step max\|float32 oracle\|ratio of the rms" gpurun_out/r03/pytest4.log | head -20
set -e
python3 scripts/bench_wgrad_bf16.py 32 1 4 8 > gpurun_out/r03/wgrad_bf16_new.txt 2>&1; cat gpurun_out/r03/wgrad_bf16_new.txt | grep -v amdgpu.ids
PIVP_WGB_KERNEL=5 python3 scripts/bench_wgrad_bf16.py 32 1 > gpurun_out/r03/wgrad_bf16_old.txt 2>&1; cat gpurun_out/r03/wgrad_bf16_old.txt | grep -v amdgpu.ids
B="--precision bf16 --mode train --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
for cfg in "new8:" "new4:PIVP_WGRAD_BATCH=4" "new1:PIVP_WGRAD_BATCH=1" "old:PIVP_WGB_KERNEL=5 PIVP_WGRAD_BATCH=1" "new8_half:PIVP_WGB_SLOTS=128"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs python3 bench.py $B > gpurun_out/r03/bf16train_$tag.json 2> gpurun_out/r03/bf16train_$tag.err
  python3 -c "import json;d=json.load(open('gpurun_out/r03/bf16train_$tag.json'));print('$tag', d['ms_per_step'], 'ms per bf16 train step, loss', d['config']['loss'])"
done
python3 bench.py --precision bf16 --no-train --no-cpu-baseline --batch 256 --steps 10 --warmup 3 > gpurun_out/r03/bf16_b256.json 2> gpurun_out/r03/bf16_b256.err
python3 -c "import json;d=json.load(open('gpurun_out/r03/bf16_b256.json'));print('bf16 rollout B=256', d['ms_per_step'], 'ms', d['roofline']['frac'], d['roofline']['per_layer_tflops'])"
