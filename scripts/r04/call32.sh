#!/bin/bash
# round 4, call 32: the whole GPU suite, smoke(), and the driver's default command on the current tree
set -o pipefail
o=gpurun_out/r04/c32
mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests.txt 2>&1 || { tail -60 $o/gpu_tests.txt; exit 1; }
tail -2 $o/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 && \
/usr/bin/time -v python bench.py > $o/bench.json 2> $o/bench.err; grep "Elapsed (wall" $o/bench.err; tail -1 $o/bench.json | cut -c1-400
