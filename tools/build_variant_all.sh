#!/bin/bash
# Build a diagnostic variant of the WHOLE library with extra flags: tools/build_variant_all.sh NAME "-DFLAG ..."
#   -> stroke-prediction_amd/lib/variants/NAME.so   (run with SP_LIB_PATH=<that file>)
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2
L=stroke-prediction_amd/lib; O=$L/variants/${NAME}_obj; mkdir -p $O
for f in stroke-prediction_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c $f -o $O/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/variants/$NAME.so $O/*.o -ldl
rm -rf $O
echo built $L/variants/$NAME.so
