set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_round5.py -k "maxpool or folded_launches" -x -q 2>&1 | tail -2
for v in "SP_ZM_VARIANT=1" "" "SP_ZM_VARIANT=1" ""; do
  echo "== $v"; env $v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('loss'))"
done
