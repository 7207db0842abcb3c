#!/bin/bash
# Sanitizer build of the C ABI's HOST code (SURVEY 5 "race detection / sanitizers"): every source compiled with
# AddressSanitizer + UndefinedBehaviorSanitizer on the host side only (-Xarch_host: the gfx950 device code is built as always;
# GPU ASan / XNACK runs are not available on the GPU pool) -> stroke-prediction_amd/lib/variants/libstroke_amd_asan.so.
# CPU box only:   tools/build_asan.sh && tools/run_asan_tests.sh
set -e
cd "$(dirname "$0")/.."
L=stroke-prediction_amd/lib; mkdir -p $L/variants/asan_obj
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -g"
pids=()
for f in stroke-prediction_amd/csrc/*.hip; do
  o=$L/variants/asan_obj/$(basename $f).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ stroke-prediction_amd/csrc/sp_common.h -nt $o ] || [ include/stroke_amd.h -nt $o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC $SAN -c $f -o $o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fsanitize=undefined -shared-libsan -o $L/variants/libstroke_amd_asan.so $L/variants/asan_obj/*.o -ldl
echo built $L/variants/libstroke_amd_asan.so
