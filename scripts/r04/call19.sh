#!/bin/bash
# round 4, call 19: the round's profile artefacts on the current tree: whole GPU suite, smoke(), and the config
# presets 3 / 4 / 5 (bench line + kernel statistics each)
set -o pipefail
o=gpurun_out/r04/c19
mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $o/gpu_tests.txt 2>&1 || { tail -60 $o/gpu_tests.txt; exit 1; }
tail -2 $o/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 && \
for c in 3 4 5; do
  timeout -k 10 400 python3 bench.py --config $c --no-cpu-baseline > $o/config${c}_bench.json 2> $o/config${c}_bench.err || { tail -5 $o/config${c}_bench.err; exit 1; }
  tail -1 $o/config${c}_bench.json | cut -c1-500
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt_c$c -o c$c -- python3 bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-train > $o/kt_c$c.log 2>&1 || { tail -5 $o/kt_c$c.log; exit 1; }
  rm -f $o/kt_c$c/*kernel_trace.csv
done
