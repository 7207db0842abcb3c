#!/bin/bash
# Build a diagnostic variant of the library: tools/build_variant.sh NAME "-DFLAG ..."  -> stroke-prediction_amd/lib/variants/NAME.so
# (only sp_conv_dma.hip / the listed file is recompiled with the flags; the other objects come from lib/obj)
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2; FILE=${3:-sp_conv_dma.hip}
L=stroke-prediction_amd/lib; mkdir -p $L/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c stroke-prediction_amd/csrc/$FILE -o $L/variants/$NAME.o
OBJS=$(ls $L/obj/*.o | grep -v "/$FILE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/variants/$NAME.so $OBJS $L/variants/$NAME.o
rm -f $L/variants/$NAME.o
echo built $L/variants/$NAME.so
