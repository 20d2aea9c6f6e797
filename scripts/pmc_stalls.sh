#!/bin/bash
# Wave-stall counters of the train step's kernels (single stream, so launches do not share CUs): two --pmc passes, per-kernel means.
# bash scripts/pmc_stalls.sh <tag>  ->  gpurun_out/stalls_<tag>/summary.txt
set -e
tag=${1:-cur}
out=gpurun_out/stalls_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PIVP_SIDE_STREAM=0
# BENCH_ARGS selects the workload (default: the fp32 train step); e.g. BENCH_ARGS="--precision bf16 --mode train" for config 3's kernels
R="${BENCH_ARGS:---mode train} --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/a -o a -- python3 bench.py $R > $out/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $out/b -o b -- python3 bench.py $R > $out/b.log 2>&1
python3 scripts/pmc_stalls.py $out/a $out/b > $out/summary.txt
find $out -name '*counter_collection.csv' -delete
cat $out/summary.txt
