#!/bin/bash
# round 4, call 37: whole GPU suite + smoke + default bench + configs 4 / 5 (each line carries the split modes' rollout and train objects) on the final tree
set -o pipefail
o=gpurun_out/r04/c37
mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests.txt 2>&1 || { tail -60 $o/gpu_tests.txt; exit 1; }
tail -2 $o/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 && \
timeout -k 10 600 python bench.py > $o/bench.json 2> $o/bench.err && \
timeout -k 10 600 python bench.py --config 4 --no-cpu-baseline > $o/config4_bench.json 2> $o/config4.err && \
timeout -k 10 900 python bench.py --config 5 --no-cpu-baseline > $o/config5_bench.json 2> $o/config5.err && \
python - <<'EOF2'
import json
for f in ('bench', 'config4_bench', 'config5_bench'):
    d = json.loads(open('gpurun_out/r04/c37/%s.json' % f).read().strip().splitlines()[-1])
    print(f, 'rollout', d['ms_per_step'], {k: d[k]['ms_per_step'] for k in d if k.startswith('rollout_') or k.startswith('train')})
EOF2
