set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py -k "two_dense or folded_launches" tests/test_gpu_unet.py -x -q > gpurun_out/t5e.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/t5e.log | cut -c1-200
for v in "SP_SPLIT_G1=0" "" "SP_SPLIT_G1=0" ""; do
  echo "== $v"; env $v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('loss'))"
done
