#!/bin/bash
# round 3, call 16: the whole GPU suite, smoke() and the default bench line on the final tree
set -o pipefail
o=gpurun_out/r03/final
mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests.txt 2>&1 || { tail -40 $o/gpu_tests.txt; exit 1; }
tail -2 $o/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 && \
timeout -k 10 600 python bench.py > $o/bench.json 2> $o/bench.err && tail -1 $o/bench.json | cut -c1-1500
