#!/bin/bash
# Timing-only ablations of the bf16 ConvLSTM kernel (results are wrong by construction): rebuilds convlstm_bf16.o with
# -DPIVP_BF16_ABL=n, relinks, runs scripts/bench_lstm_layers.py, and restores the real library at the end.
set -e
P=physical-interaction-video-prediction_amd
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
OBJS=$(ls $P/csrc/*.o)
export PIVP_BENCH_BF16=1
run() { timeout -k 10 200 python scripts/bench_lstm_layers.py ${1:-32} 20; }
echo "== full"; run $1
for a in ${ABLS:-1 2 3}; do
  /opt/rocm/bin/hipcc $FL -DPIVP_BF16_ABL=$a -c $P/csrc/convlstm_bf16.hip -o $P/csrc/convlstm_bf16.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libpivp_hip.so $OBJS
  echo "== ablation $a"; run $1
done
/opt/rocm/bin/hipcc $FL -c $P/csrc/convlstm_bf16.hip -o $P/csrc/convlstm_bf16.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libpivp_hip.so $OBJS
