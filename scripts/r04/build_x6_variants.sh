#!/bin/bash
# timing-only variants of the three-piece ConvLSTM kernel (PIVP_X6_ABL), built here and shipped next to the library:
#   physical-interaction-video-prediction_amd/variants/libpivp_hip_x6abl<n>.so      (scripts/bench_lstm_layers.py takes PIVP_BENCH_LIB)
set -e
pkg=physical-interaction-video-prediction_amd
mkdir -p $pkg/variants
python $pkg/build.py > /dev/null
dig=$(python -c "import sys; sys.path.insert(0, '$pkg'); import _digest; print(_digest.source_digest())")
for n in "$@"; do
  def="-DPIVP_X6_ABL=$n"
  [ "$n" = stamps ] && def="-DPIVP_BF16_STAMPS"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $def -c $pkg/csrc/convlstm_bf16.hip -o /tmp/convlstm_bf16_abl$n.o &
done
wait
for n in "$@"; do
  objs=$(ls $pkg/csrc/*.o | grep -v convlstm_bf16.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/variants/libpivp_hip_x6abl$n.so $objs /tmp/convlstm_bf16_abl$n.o
done
ls -la $pkg/variants
