#!/bin/bash
# round 4, call 44: the options kept from this round's measured nulls still pass their tests on the final tree
set -o pipefail
o=gpurun_out/r04/c44
mkdir -p $o
run() { echo "== $1" | tee -a $o/summary.txt; shift; env "$@" 2>&1 | tail -1 | tee -a $o/summary.txt; }
run "PIVP_LN_BWD=2 gradients" PIVP_LN_BWD=2 timeout -k 10 500 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward_ops.py -x -q
run "PIVP_LN_BWD=1 gradients" PIVP_LN_BWD=1 timeout -k 10 500 python -m pytest tests/test_gpu_train.py -x -q
run "PIVP_FRAME_HEAD=0 model + trained" PIVP_FRAME_HEAD=0 timeout -k 10 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_trained.py -x -q
run "PIVP_LN_FOLD_LSTM=0 split modes" PIVP_LN_FOLD_LSTM=0 timeout -k 10 500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trained.py -x -q -k "fp16x3 or x6 or split"
run "PIVP_LSTM_SQ=1 model" PIVP_LSTM_SQ=1 timeout -k 10 500 python -m pytest tests/test_gpu_model.py -x -q
run "PIVP_WGRAD_OCC=3 PIVP_WGRAD_BATCH=4 train" PIVP_WGRAD_OCC=3 PIVP_WGRAD_BATCH=4 timeout -k 10 500 python -m pytest tests/test_gpu_train.py -x -q
run "PIVP_BF16_DEPTH=0 PIVP_ENC4_PARTIALS=1 bf16 + model" PIVP_BF16_DEPTH=0 PIVP_ENC4_PARTIALS=1 timeout -k 10 500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_model.py -x -q
