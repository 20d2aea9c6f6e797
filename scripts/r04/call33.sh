#!/bin/bash
# round 4, call 33: scripts/profile_round.sh on the final tree: profiles/r04/v3_*
set -o pipefail
o=gpurun_out/r04/c33
mkdir -p $o
t0=$(date +%s); timeout -k 10 600 python bench.py > $o/bench_default.json 2> $o/bench_default.err || { tail -5 $o/bench_default.err; exit 1; }; echo "default bench.py: $(( $(date +%s) - t0 )) s"
timeout -k 10 1000 bash scripts/profile_round.sh v3 > $o/profile_round.log 2>&1 || { tail -30 $o/profile_round.log; exit 1; }
tail -8 $o/profile_round.log | cut -c1-300
