set -o pipefail
for v in "SP_PLAN_ROWS_WIDE=4,4,4" "SP_PLAN_ROWS_WIDE=2,2,4" "SP_PLAN_ROWS_WIDE=2,1,8" "SP_PLAN_ROWS_WIDE=4,4,4" "SP_PLAN_ROWS_WIDE=2,2,4"; do
  echo "== $v"; env "$v" timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --layers > gpurun_out/b.out 2> gpurun_out/b.err; grep -E "conv_igemm  (32->64|64->64|64->32)" gpurun_out/b.err | cut -c1-75 | tr -s ' ' | tr '\n' '|'; echo; python -c "import json; d=json.loads(open('gpurun_out/b.out').read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], d['config'].get('loss'))"
done
