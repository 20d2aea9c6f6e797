#!/bin/bash
# round 4, call 27: scripts/profile_round.sh on the current tree (with the three-piece mode's legs): profiles/r04/v2_*
set -o pipefail
o=gpurun_out/r04/c27
mkdir -p $o
timeout -k 10 1150 bash scripts/profile_round.sh v2 > $o/profile_round.log 2>&1 || { tail -30 $o/profile_round.log; exit 1; }
tail -6 $o/profile_round.log | cut -c1-400
