set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_round5.py tests/test_gpu_round3.py -x -q 2>&1 | tail -3
for v in "SP_ZM_PSER=" "X=1"; do
  echo "== x3 $v"; env "$v" timeout -k 10 200 python bench.py --dtype bf16x3 --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], d['config'].get('loss'))"
  echo "== f16x3 $v"; env "$v" timeout -k 10 200 python bench.py --dtype f16x3 --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], d['config'].get('loss'))"
done
