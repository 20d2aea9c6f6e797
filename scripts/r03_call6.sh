#!/bin/bash
# Round 3, GPU call 6: enc0 data gradient by parity class, coalesced mask-softmax backward, double-buffered heads backward: parity + train steps; per-block bf16 stamps.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > gpurun_out/r03/pytest6.log 2>&1 || echo "TESTS FAILED (see gpurun_out/r03/pytest6.log)"
tail -5 gpurun_out/r03/pytest6.log
set -e
python3 bench.py --no-cpu-baseline > gpurun_out/r03/bench6.json 2> gpurun_out/r03/bench6.err
python3 -c "
import json
d=json.load(open('gpurun_out/r03/bench6.json'))
print('rollout', d['ms_per_step'], 'frac', d['roofline']['frac'], 'train', d['train']['ms_per_step'], 'train_bf16', d['train_bf16']['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03/trace_train_single6
mkdir -p $out
PIVP_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o t -- python3 bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $out/log 2>&1
python3 scripts/queue_breakdown.py $out/kt/t_kernel_trace.csv 4 0.4 > $out/queues.txt 2>&1 || true
rm -f $out/kt/*kernel_trace.csv
sed -n 1,45p $out/queues.txt
PIVP_EXTRA_FLAGS="-DPIVP_BF16_STAMPS" python3 physical-interaction-video-prediction_amd/build.py --force > gpurun_out/r03/build_stamps6.log 2>&1
python3 scripts/bf16_stamps.py > gpurun_out/r03/bf16_stamps_blocks.txt 2>&1
cat gpurun_out/r03/bf16_stamps_blocks.txt | grep -v amdgpu
