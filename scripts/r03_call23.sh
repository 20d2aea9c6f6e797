#!/bin/bash
# round 3, call 23: the STP model's train step: bench line and kernel table (where does composite_bwd_stp_kernel stand?)
set -o pipefail
o=$GRAFT_REPO_ROOT/gpurun_out/r03/stp_train
mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --model STP --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $o/bench.json 2> $o/bench.err || { tail $o/bench.err; exit 1; }
python3 -c "import json;d=json.loads(open('$o/bench.json').read().strip().splitlines()[-1]);print('STP train ms_per_step', d['ms_per_step'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt -o t -- python3 bench.py --model STP --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $o/kt.log 2>&1 || { tail -5 $o/kt.log; exit 1; }
f=$(find $o/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $o/stp_train_kernel_stats.csv
find $o/kt -name "*.csv" ! -name "*kernel_stats.csv" -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$o/stp_train_kernel_stats.csv')))
for r in rows[:14]:
    print('%-70s calls %5s avg %8.1f us  %5.1f%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
