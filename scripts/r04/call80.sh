#!/bin/bash
# round 4, call 80: the final tree: whole GPU suite, smoke(), the driver's default command
set -o pipefail
o=gpurun_out/r04/c80
mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests.txt 2>&1 || { tail -60 $o/gpu_tests.txt; exit 1; }
tail -2 $o/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 && \
t0=$(date +%s) && timeout -k 10 600 python bench.py > $o/bench.json 2> $o/bench.err && echo "default bench.py: $(( $(date +%s) - t0 )) s" && python - <<'EOF2'
import json
d = json.loads(open('gpurun_out/r04/c80/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], {k: d[k]['ms_per_step'] for k in d if k.startswith('rollout_') or k.startswith('train')})
EOF2
