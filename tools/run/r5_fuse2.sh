set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_kernels.py -x -q > gpurun_out/t5.log 2>&1; echo "round5+kernels rc $?"; tail -4 gpurun_out/t5.log
for v in "SP_FUSE_BN_FINALIZE=0 SP_FUSE_DZ=0" "SP_FUSE_DZ=0" "" "SP_FUSE_BN_FINALIZE=0 SP_FUSE_DZ=0" "SP_FUSE_DZ=0" ""; do
  echo "== $v"; env $v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('loss'))"
done
