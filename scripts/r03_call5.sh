#!/bin/bash
# Round 3, GPU call 5: whole GPU suite (new op tests, trained fixtures, batched bf16 weight gradient), bf16 train step with the kernel chosen by batch size, one-stream bf16 trace.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r03/pytest5.log 2>&1 || echo "TESTS FAILED (see gpurun_out/r03/pytest5.log)"
tail -6 gpurun_out/r03/pytest5.log
python3 -m pytest tests/test_gpu_trained.py -m gpu -q -s 2>&1 | grep -h "per-step max\|float32 oracle\|ratio of the rms" | cut -c1-400
set -e
B="--precision bf16 --mode train --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
for cfg in "default:" "old:PIVP_WGB_KERNEL=5 PIVP_WGRAD_BATCH=1" "b4:PIVP_WGRAD_BATCH=4" "noside:PIVP_SIDE_STREAM=0" "noside_old:PIVP_SIDE_STREAM=0 PIVP_WGB_KERNEL=5 PIVP_WGRAD_BATCH=1"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs python3 bench.py $B > gpurun_out/r03/bf16train5_$tag.json 2> gpurun_out/r03/bf16train5_$tag.err
  python3 -c "import json;d=json.load(open('gpurun_out/r03/bf16train5_$tag.json'));print('$tag', d['ms_per_step'], 'ms per bf16 train step')"
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03/trace_bf16_single
mkdir -p $out
PIVP_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o t -- python3 bench.py --precision bf16 --mode train --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $out/log 2>&1
python3 scripts/queue_breakdown.py $out/kt/t_kernel_trace.csv 4 0.4 > $out/queues.txt 2>&1 || true
rm -f $out/kt/*kernel_trace.csv
sed -n 1,40p $out/queues.txt
