#!/bin/bash
# round 3, call 21: operand-value dependence of the fp32 weight-gradient / data-gradient kernels (lstm7 cell backward, B = 32)
set -o pipefail
o=$GRAFT_REPO_ROOT/gpurun_out/r03/wgrad_data
mkdir -p $o
cd /tmp && export TMPDIR=/tmp
for v in "random random" "random zero" "zero random" "zero zero"; do
  set -- $v
  d=$o/x$1_dg$2
  (cd $GRAFT_REPO_ROOT && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 scripts/wgrad_data_dependence.py --x $1 --dg $2 > $d.log 2>&1) || { tail -5 $d.log; exit 1; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || { echo "no stats in $d"; exit 1; }
  python3 - "$f" "$1" "$2" <<'PY' | tee -a $o/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    n = r['Name']
    for k in ('wgrad5x5_kernel', 'igemm_f32_kernel', 'lstm_gates_bwd_kernel'):
        if 'pivp::' + k in n:
            out.append('%s %.1f us' % (k, float(r['AverageNs']) / 1e3))
print('x %-6s dG %-6s: %s' % (sys.argv[2], sys.argv[3], ', '.join(out)))
PY
  find $d -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
