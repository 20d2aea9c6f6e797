#!/bin/bash
# round 4, call 75: the config presets on the final tree (bench lines with the split modes' objects)
set -o pipefail
o=gpurun_out/r04/c75
mkdir -p $o
for c in 3 4 5; do
  timeout -k 10 500 python bench.py --config $c --no-cpu-baseline > $o/config${c}_bench.json 2> $o/config${c}.err || { tail -5 $o/config${c}.err; exit 1; }
  python - <<EOF2
import json
d = json.loads(open('$o/config${c}_bench.json').read().strip().splitlines()[-1])
print('config $c', d['ms_per_step'], d['value'], d['unit'], {k: d[k]['ms_per_step'] for k in d if isinstance(d[k], dict) and 'ms_per_step' in d[k]})
EOF2
done
