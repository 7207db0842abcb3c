#!/bin/bash
# Host-logic tests of the C ABI (export set, planner tables, argument validation, RCCL id plumbing) against the sanitizer
# build.  CPU box only -- never on the GPU pool.
cd "$(dirname "$0")/.."
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
SP_LIB_PATH=$PWD/stroke-prediction_amd/lib/variants/libstroke_amd_asan.so LD_PRELOAD=$RT \
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:detect_odr_violation=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_cabi.py -x -q -p no:cacheprovider "$@"
