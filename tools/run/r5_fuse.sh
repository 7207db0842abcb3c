set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/t5.log 2>&1; echo "round5 rc $?"; tail -15 gpurun_out/t5.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round5.py > gpurun_out/tall.log 2>&1; echo "all rc $?"; tail -8 gpurun_out/tall.log
for v in "SP_FUSE_BN_FINALIZE=0 SP_FUSE_DZ=0 SP_ZM_TILE=16" "SP_FUSE_BN_FINALIZE=0 SP_FUSE_DZ=0" "SP_FUSE_DZ=0" ""; do
  echo "== $v"; env $v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('loss'))"
done
