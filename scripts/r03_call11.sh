#!/bin/bash
# Round 3, GPU call 11: heads kernel with gamma / beta requested up front, transposed-conv epilogue through LDS (float4 rows): parity + bench.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > gpurun_out/r03/pytest11.log 2>&1 || echo "TESTS FAILED (see gpurun_out/r03/pytest11.log)"
tail -4 gpurun_out/r03/pytest11.log
set -e
python3 bench.py --no-cpu-baseline > gpurun_out/r03/bench11.json 2> gpurun_out/r03/bench11.err
python3 -c "
import json
d=json.load(open('gpurun_out/r03/bench11.json'))
print('rollout', d['ms_per_step'], 'frac', d['roofline']['frac'], 'share', d['roofline']['share_of_step_time'], 'train', d['train']['ms_per_step'], 'train_bf16', d['train_bf16']['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/kt11 -o rollout -- python3 bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-train > gpurun_out/r03/kt11.log 2>&1
rm -f gpurun_out/r03/kt11/*kernel_trace.csv
grep -h "deconv3x3s2_tile\|heads_1x1" gpurun_out/r03/kt11/rollout_kernel_stats.csv | cut -c1-60,150-230
