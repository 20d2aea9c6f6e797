#!/bin/bash
# round 3, call 25: the other heads: STP / DNA rollout and train step (bench lines), and the kernel table of the DNA train step
set -o pipefail
o=$GRAFT_REPO_ROOT/gpurun_out/r03/variants
mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in CDNA STP DNA; do
  python3 bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $o/bench_$m.json 2> $o/err.txt || { tail $o/err.txt; exit 1; }
  python3 -c "import json;d=json.loads(open('$o/bench_$m.json').read().strip().splitlines()[-1]);print('$m: rollout %.3f ms, train %.3f ms, train_bf16 %.3f ms' % (d['ms_per_step'], d['train']['ms_per_step'], d['train_bf16']['ms_per_step']))"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt -o t -- python3 bench.py --model DNA --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $o/kt.log 2>&1 || { tail -5 $o/kt.log; exit 1; }
f=$(find $o/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $o/dna_train_kernel_stats.csv
find $o/kt -name "*.csv" ! -name "*kernel_stats.csv" -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$o/dna_train_kernel_stats.csv')))
for r in rows[:40]:
    n=r['Name']
    if any(k in n for k in ('composite','heads','dna','softmax','enc0_')):
        print('%-70s calls %5s avg %8.1f us  %5.1f%%' % (n[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
