set -e
out=gpurun_out/prof_bf16train
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o train_bf16 -- python3 bench.py --precision bf16 --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $out/log 2>&1
rm -f $out/kt/*kernel_trace.csv
head -22 $out/kt/train_bf16_kernel_stats.csv | cut -c1-150
