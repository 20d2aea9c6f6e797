#!/bin/bash
# round 4, call 36: the fragment-major pack with one index decode per 8 x pieces elements, absmax with 16-byte loads: split-mode tests, kernel statistics
set -o pipefail
o=gpurun_out/r04/c36
mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py -x -q -k "fp16x3 or x6" > $o/tests.txt 2>&1 || { tail -40 $o/tests.txt; exit 1; }
tail -1 $o/tests.txt
for p in fp16x3 bf16x6; do
  timeout -k 10 200 python3 bench.py --precision $p --no-cpu-baseline --no-train > $o/bench_$p.json 2> $o/err.txt || { tail -5 $o/err.txt; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt_$p -o r -- python3 bench.py --precision $p --steps 7 --warmup 2 --no-cpu-baseline --no-train --no-roofline > $o/kt_$p.log 2>&1 || { tail -5 $o/kt_$p.log; exit 1; }
  rm -f $o/kt_$p/*kernel_trace.csv
done
python3 - <<'EOF2'
import json, csv
for p in ('fp16x3', 'bf16x6'):
    d = json.loads(open('gpurun_out/r04/c36/bench_%s.json' % p).read().strip().splitlines()[-1])
    print(p, d['ms_per_step'])
    rows = list(csv.DictReader(open('gpurun_out/r04/c36/kt_%s/r_kernel_stats.csv' % p)))
    for r in rows[:16]:
        print('   %-70s %5d calls  %8.1f us each  %8.1f us per rollout' % (r['Name'][:70], int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3 / 9))
EOF2
