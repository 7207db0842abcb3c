set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/t5.log 2>&1; echo "tests rc $?"; tail -12 gpurun_out/t5.log | cut -c1-200
for v in "SP_FUSE_POOL=0" "" "SP_FUSE_POOL=0" "" "SP_FUSE_POOL=0 SP_FUSE_BN_FINALIZE=0 SP_FUSE_DZ=0 SP_ZM_TILE=16"; do
  echo "== $v"; env $v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('loss'))"
done
for v in "SP_FUSE_POOL=0 SP_FUSE_BN_FINALIZE=0 SP_FUSE_DZ=0 SP_ZM_TILE=16" ""; do
  echo "== x3 $v"; env $v timeout -k 10 200 python bench.py --dtype bf16x3 --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('loss'))"
done
