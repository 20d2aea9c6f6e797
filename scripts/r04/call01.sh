#!/bin/bash
# round 4, call 1: start-of-round state on a fresh box: the GPU suite (with this round's new tests: stale-library guard, rs_ag kernels,
# DeviceFeeder loop), smoke(), the default bench line, and the config presets 3 / 4.
set -o pipefail
o=gpurun_out/r04/c01
mkdir -p $o
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $o/gpu_tests.txt 2>&1 || { tail -60 $o/gpu_tests.txt; exit 1; }
tail -2 $o/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 && \
timeout -k 10 600 python bench.py > $o/bench.json 2> $o/bench.err && tail -1 $o/bench.json | cut -c1-2500 && \
timeout -k 10 300 python bench.py --config 4 --no-cpu-baseline --no-train > $o/bench_c4.json 2> $o/bench_c4.err && tail -1 $o/bench_c4.json | cut -c1-600 && \
timeout -k 10 300 python bench.py --config 3 --no-cpu-baseline > $o/bench_c3.json 2> $o/bench_c3.err && tail -1 $o/bench_c3.json | cut -c1-600
