#!/bin/bash
# round 4, call 12: LayerNorm-backward sums from the producers' epilogues + parameter gradients inside the gate backward (PIVP_LN_BWD 2 / 1 / 0)
set -o pipefail
o=gpurun_out/r04/c12
mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_backward_ops.py tests/test_gpu_train.py tests/test_gpu_configs.py tests/test_gpu_bf16.py -m gpu -x -q > $o/tests.txt 2>&1 || { tail -80 $o/tests.txt; exit 1; }
tail -2 $o/tests.txt
for rep in 1 2; do
  for m in 0 1 2; do
    PIVP_LN_BWD=$m timeout -k 10 200 python bench.py --mode train --no-cpu-baseline --no-roofline --steps 20 > $o/train_m${m}_$rep.json 2>> $o/err.txt || exit 1
    echo "fp32 train PIVP_LN_BWD=$m rep $rep: $(python -c "import json; print(json.loads(open('$o/train_m${m}_$rep.json').read().splitlines()[-1])['ms_per_step'])") ms"
    PIVP_LN_BWD=$m timeout -k 10 200 python bench.py --mode train --precision bf16 --no-cpu-baseline --no-roofline --steps 20 > $o/train16_m${m}_$rep.json 2>> $o/err.txt || exit 1
    echo "bf16 train PIVP_LN_BWD=$m rep $rep: $(python -c "import json; print(json.loads(open('$o/train16_m${m}_$rep.json').read().splitlines()[-1])['ms_per_step'])") ms"
  done
done
