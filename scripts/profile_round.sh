#!/bin/bash
# One GPU-box call that produces every profile artefact kept under profiles/<round>/ (run from the repo root through gpurun):
#   bash scripts/profile_round.sh <tag>        -> gpurun_out/prof_<tag>/...
# rocprofv3 gets the program itself after `--` (python3 bench.py ...), and counters are collected in their own --pmc passes.
set -e -o pipefail
tag=${1:-cur}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NB="--no-cpu-baseline"
python3 bench.py > $out/bench.json 2> $out/bench.err                                   # the driver's command: rollout + train leg + cpu baseline
python3 bench.py --mode train --steps 10 --warmup 3 $NB > $out/bench_train.json 2> $out/bench_train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o rollout -- python3 bench.py --steps 7 --warmup 2 $NB --no-train --no-bf16x6 > $out/kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/train -o train -- python3 bench.py --mode train --steps 4 --warmup 1 $NB --no-roofline > $out/train.log 2>&1
python3 scripts/overlap_report.py $out/train/train_kernel_trace.csv > $out/train_overlap.txt 2>&1 || true
python3 bench.py --precision bf16 $NB > $out/bench_bf16.json 2> $out/bench_bf16.err
python3 bench.py --precision bf16 --mode train --steps 10 --warmup 3 $NB --no-roofline > $out/bench_bf16_train.json 2> $out/bench_bf16_train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_bf16 -o rollout_bf16 -- python3 bench.py --precision bf16 --steps 7 --warmup 2 $NB --no-train > $out/kt_bf16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_bf16 -o train_bf16 -- python3 bench.py --precision bf16 --mode train --steps 4 --warmup 1 $NB --no-roofline > $out/train_bf16.log 2>&1
# the three-piece mode (fp32-grade on the bf16 matrix cores): rollout and train step, kernel statistics, two-queue overlap of the train step
python3 bench.py --precision bf16x6 $NB --no-train > $out/bench_x6.json 2> $out/bench_x6.err
python3 bench.py --precision bf16x6 --mode train --steps 10 --warmup 3 $NB --no-roofline > $out/bench_x6_train.json 2> $out/bench_x6_train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_x6 -o rollout_x6 -- python3 bench.py --precision bf16x6 --steps 7 --warmup 2 $NB --no-train > $out/kt_x6.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_x6 -o train_x6 -- python3 bench.py --precision bf16x6 --mode train --steps 4 --warmup 1 $NB --no-roofline > $out/train_x6.log 2>&1
python3 scripts/overlap_report.py $out/train_x6/train_x6_kernel_trace.csv > $out/train_x6_overlap.txt 2>&1 || true
python3 scripts/queue_breakdown.py $out/train_x6/train_x6_kernel_trace.csv > $out/train_x6_queues.txt 2>&1 || true
# two fp16 pieces per operand (forward gate convolutions, their data and weight gradients)
python3 bench.py --precision fp16x3 $NB --no-train > $out/bench_fp16x3.json 2> $out/bench_fp16x3.err
python3 bench.py --precision fp16x3 --mode train --steps 10 --warmup 3 $NB --no-roofline > $out/bench_fp16x3_train.json 2> $out/bench_fp16x3_train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_fp16x3 -o rollout_fp16x3 -- python3 bench.py --precision fp16x3 --steps 7 --warmup 2 $NB --no-train > $out/kt_fp16x3.log 2>&1
rm -f $out/kt_fp16x3/*kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_fp16x3 -o train_fp16x3 -- python3 bench.py --precision fp16x3 --mode train --steps 4 --warmup 1 $NB --no-roofline > $out/train_fp16x3.log 2>&1
python3 scripts/overlap_report.py $out/train_fp16x3/train_fp16x3_kernel_trace.csv > $out/train_fp16x3_overlap.txt 2>&1 || true
python3 scripts/queue_breakdown.py $out/train_fp16x3/train_fp16x3_kernel_trace.csv > $out/train_fp16x3_queues.txt 2>&1 || true
rm -f $out/train_fp16x3/*kernel_trace.csv
R="--steps 2 --warmup 1 $NB --no-roofline --no-train --no-bf16x6"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 bench.py $R > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 bench.py $R > $out/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -o m -- python3 bench.py $R > $out/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_bf16 -o f -- python3 bench.py --precision bf16 $R > $out/fetch_bf16.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write_bf16 -o w -- python3 bench.py --precision bf16 $R > $out/write_bf16.log 2>&1
python3 scripts/pmc_summary.py $out/fetch $out/write $out/pmc "bench.py --steps 2 --warmup 1 (config 2: CDNA B=32 T=10 64x64)"
python3 scripts/pmc_summary.py --bf16 $out/fetch_bf16 $out/write_bf16 $out/pmc "bench.py --precision bf16 --steps 2 --warmup 1 (CDNA B=32 T=10 64x64)"
python3 scripts/pmc_summary.py --mfma $out/mfma $out/pmc/pmc_mfma_busy_summary.csv
# keep the merge small: the raw per-dispatch traces are not needed back
rm -f $out/kt/*kernel_trace.csv $out/train/*kernel_trace.csv $out/kt_bf16/*kernel_trace.csv $out/train_bf16/*kernel_trace.csv $out/kt_x6/*kernel_trace.csv $out/train_x6/*kernel_trace.csv
find $out/fetch $out/write $out/mfma $out/fetch_bf16 $out/write_bf16 -name '*counter_collection.csv' -delete
cat $out/bench.json; cat $out/bench_train.json; cat $out/bench_bf16.json; cat $out/bench_bf16_train.json; cat $out/bench_x6.json; cat $out/bench_x6_train.json; cat $out/bench_fp16x3.json; cat $out/bench_fp16x3_train.json
