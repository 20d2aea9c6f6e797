#!/bin/bash
# round 4, call 42: the norms of hidden1 / hidden3 applied while lstm2 / lstm4 stage their patch (split modes, inference): tests, bench
set -o pipefail
o=gpurun_out/r04/c42
mkdir -p $o
timeout -k 10 800 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py tests/test_gpu_model.py -x -q -k "fp16x3 or x6 or split or golden or packs" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
for f in 1 0; do
  PIVP_LN_FOLD_LSTM=$f timeout -k 10 400 python bench.py --no-cpu-baseline --no-train > $o/bench_fold$f.json 2> $o/bench.err || { tail -5 $o/bench.err; exit 1; }
done
python - <<'EOF2'
import json
for f in (1, 0):
    d = json.loads(open('gpurun_out/r04/c42/bench_fold%d.json' % f).read().strip().splitlines()[-1])
    print('fold', f, 'rollout', d['ms_per_step'], {k: d[k]['ms_per_step'] for k in d if k.startswith('rollout_')})
EOF2
