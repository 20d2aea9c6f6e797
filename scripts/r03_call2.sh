#!/bin/bash
# Round 3, GPU call 2: parity suite on the edited kernels, A/B of the priority staircase, bf16 after the bank-conflict fix, better-trained fixture weights.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest2.log 2>&1 || echo "TESTS FAILED (see gpurun_out/r03/pytest2.log)"
tail -5 gpurun_out/r03/pytest2.log
set -e
python3 bench.py --no-cpu-baseline > gpurun_out/r03/bench1.json 2> gpurun_out/r03/bench1.err
PIVP_PRIO_STAIRS=0 python3 bench.py --no-cpu-baseline > gpurun_out/r03/bench1_nostairs.json 2> gpurun_out/r03/bench1_nostairs.err
python3 bench.py --no-cpu-baseline --precision bf16 --no-train > gpurun_out/r03/bench1_bf16.json 2> gpurun_out/r03/bench1_bf16.err
python3 - <<'PY'
import json
for n in ('bench1', 'bench1_nostairs', 'bench1_bf16'):
    d = json.load(open('gpurun_out/r03/%s.json' % n))
    r = d['roofline']
    print(n, 'ms', d['ms_per_step'], 'frac', r['frac'], r['per_layer_tflops'], 'train', (d.get('train') or {}).get('ms_per_step'), 'bf16 train', (d.get('train_bf16') or {}).get('ms_per_step'))
PY
python3 tests/golden/train_weights.py --model STP --size 64 --steps 3000 --out gpurun_out/r03/trained_stp64_q8.npz > gpurun_out/r03/train_stp64.log 2>&1
tail -6 gpurun_out/r03/train_stp64.log
python3 tests/golden/train_weights.py --model CDNA --size 128 --steps 1000 --freeze model/cdna_kerns/W --out gpurun_out/r03/trained_cdna128_q8.npz > gpurun_out/r03/train_cdna128.log 2>&1
tail -6 gpurun_out/r03/train_cdna128.log
