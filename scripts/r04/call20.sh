#!/bin/bash
# round 4, call 20: scripts/profile_round.sh on the current tree (bench lines, kernel statistics, counters: profiles/r04/v1_*)
set -o pipefail
o=gpurun_out/r04/c20
mkdir -p $o
timeout -k 10 1150 bash scripts/profile_round.sh v1 > $o/profile_round.log 2>&1 || { tail -30 $o/profile_round.log; exit 1; }
tail -4 $o/profile_round.log | cut -c1-700
