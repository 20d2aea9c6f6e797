#!/bin/bash
# round 4, call 39: weight packs kept across rollouts while the parameters are untouched: invalidation test, whole bf16 / trained suites, bench
set -o pipefail
o=gpurun_out/r04/c39
mkdir -p $o
timeout -k 10 800 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py tests/test_gpu_train.py -x -q > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
timeout -k 10 400 python bench.py --no-cpu-baseline > $o/bench.json 2> $o/bench.err && python - <<'EOF2'
import json
d = json.loads(open('gpurun_out/r04/c39/bench.json').read().strip().splitlines()[-1])
print('rollout', d['ms_per_step'], {k: d[k]['ms_per_step'] for k in d if k.startswith('rollout_') or k.startswith('train')})
EOF2
timeout -k 10 200 python bench.py --precision bf16 --no-cpu-baseline --no-train > $o/bench_bf16.json 2>> $o/bench.err && tail -1 $o/bench_bf16.json | cut -c1-200
