#!/bin/bash
# Round 3, GPU call 3: where a bf16 train step spends its two queues; in-kernel clock of the fp32 ConvLSTM loop; bf16 phase stamps; trained weights (feed-self, longer).
set -e -o pipefail
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03/trace_bf16_train
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o t -- python3 bench.py --precision bf16 --mode train --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $out/log 2>&1
python3 scripts/overlap_report.py $out/kt/t_kernel_trace.csv 0.4 > $out/overlap.txt 2>&1 || true
python3 scripts/queue_breakdown.py $out/kt/t_kernel_trace.csv 4 0.4 > $out/queues.txt 2>&1 || true
rm -f $out/kt/*kernel_trace.csv
cat $out/overlap.txt | head -8; cat $out/queues.txt
python3 tests/golden/train_weights.py --model STP --size 64 --steps 8000 --out gpurun_out/r03/trained_stp64_q8.npz > gpurun_out/r03/train_stp64.log 2>&1
tail -6 gpurun_out/r03/train_stp64.log
python3 tests/golden/train_weights.py --model CDNA --size 128 --steps 2000 --freeze model/cdna_kerns/W --out gpurun_out/r03/trained_cdna128_q8.npz > gpurun_out/r03/train_cdna128.log 2>&1
tail -6 gpurun_out/r03/train_cdna128.log
PIVP_EXTRA_FLAGS="-DPIVP_F32_STAMPS -DPIVP_BF16_STAMPS" python3 physical-interaction-video-prediction_amd/build.py --force > gpurun_out/r03/build_stamps.log 2>&1
python3 scripts/f32_stamps.py > gpurun_out/r03/f32_stamps.txt 2>&1
cat gpurun_out/r03/f32_stamps.txt
python3 scripts/bf16_stamps.py > gpurun_out/r03/bf16_stamps.txt 2>&1
cat gpurun_out/r03/bf16_stamps.txt
