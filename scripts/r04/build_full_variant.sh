#!/bin/bash
# a whole-library variant build with extra flags, shipped next to the library:  bash scripts/r04/build_full_variant.sh <name> <flags...>
#   -> physical-interaction-video-prediction_amd/variants/libpivp_hip_<name>.so   (scripts/r04/bench_with_lib.py runs bench.py on it)
set -e
name=$1; shift
pkg=physical-interaction-video-prediction_amd
mkdir -p $pkg/variants /tmp/pivp_var_$name
dig=$(python -c "import sys; sys.path.insert(0, '$pkg'); import _digest; print(_digest.source_digest())")
for f in $pkg/csrc/*.hip; do
  b=$(basename $f .hip)
  extra=""
  [ $b = pivp_c_api ] && extra="-DPIVP_BUILD_DIGEST=\"$dig\""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" $extra -c $f -o /tmp/pivp_var_$name/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/variants/libpivp_hip_$name.so /tmp/pivp_var_$name/*.o
ls -la $pkg/variants/libpivp_hip_$name.so
