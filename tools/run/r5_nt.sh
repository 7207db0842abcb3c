set -o pipefail
for v in "" "SP_LIB_PATH=$PWD/stroke-prediction_amd/lib/variants/nt.so" "" "SP_LIB_PATH=$PWD/stroke-prediction_amd/lib/variants/nt.so"; do
  echo "== $v"; env $v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('loss'))"
done
