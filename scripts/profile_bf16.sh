set -e
out=gpurun_out/prof_v13
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --precision bf16 --no-cpu-baseline > $out/bench_bf16.json 2> $out/bench_bf16.err
python3 bench.py --precision bf16 --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $out/bench_bf16_train.json 2> $out/bench_bf16_train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_bf16 -o rollout_bf16 -- python3 bench.py --precision bf16 --steps 7 --warmup 2 --no-cpu-baseline > $out/kt_bf16.log 2>&1
rm -f $out/kt_bf16/*kernel_trace.csv
cat $out/bench_bf16.json $out/bench_bf16_train.json | cut -c1-200
