#!/bin/bash
# Round 3, GPU call 9: whole GPU suite with the CDNA-64 / DNA trained fixtures, smoke(), the STP train-mode bench with its CPU baseline.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r03/pytest9.log 2>&1 || echo "TESTS FAILED (see gpurun_out/r03/pytest9.log)"
tail -4 gpurun_out/r03/pytest9.log
python3 -m pytest tests/test_gpu_trained.py -m gpu -q -s 2>&1 | grep -h "HIP \[\|gradients:\|bf16 mode on trained" | cut -c1-420
set -e
python3 -c "import __graft_entry__ as g; g.smoke()"
python3 bench.py --model STP --mode train --steps 5 --warmup 2 --cpu-seconds 4 > gpurun_out/r03/bench_stp_train.json 2> gpurun_out/r03/bench_stp_train.err
python3 -c "import json;d=json.load(open('gpurun_out/r03/bench_stp_train.json'));print('STP train', d['ms_per_step'], d['cpu_baseline'])"
