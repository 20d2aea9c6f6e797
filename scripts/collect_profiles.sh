#!/bin/bash
# Copy the summaries of a scripts/profile_round.sh pass from gpurun_out/prof_<tag>/ into profiles/<round>/ as <tag>_*:
# bash scripts/collect_profiles.sh <tag> <round>
set -e
tag=$1; rd=${2:-r02}
src=gpurun_out/prof_$tag; dst=profiles/$rd
mkdir -p $dst
cp $src/bench.json $dst/${tag}_bench.json
cp $src/bench_train.json $dst/${tag}_train_bench.json
cp $src/bench_bf16.json $dst/${tag}_bf16_bench.json
cp $src/bench_bf16_train.json $dst/${tag}_bf16_train_bench.json
cp $src/kt/rollout_kernel_stats.csv $dst/${tag}_kernel_stats.csv
cp $src/train/train_kernel_stats.csv $dst/${tag}_train_kernel_stats.csv
cp $src/kt_bf16/rollout_bf16_kernel_stats.csv $dst/${tag}_bf16_kernel_stats.csv
cp $src/train_bf16/train_bf16_kernel_stats.csv $dst/${tag}_bf16_train_kernel_stats.csv
cp $src/train_overlap.txt $dst/${tag}_train_overlap.txt
cp $src/pmc/* $dst/
ls $dst | wc -l
