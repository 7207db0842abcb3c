set -o pipefail
SP_ZM_TILE=32x16 timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round5.py -x -q 2>&1 | tail -2
for v in "SP_ZM_TILE=auto" "SP_ZM_TILE=32x16" "SP_ZM_TILE=auto" "SP_ZM_TILE=32x16"; do
  echo "== $v"; env "$v" timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --layers > gpurun_out/b.out 2> gpurun_out/b.err; grep -E "conv_igemm  16->16|conv_igemm  32->16" gpurun_out/b.err | cut -c1-75 | tr -s ' ' | tr '\n' '|'; echo; python -c "import json; d=json.loads(open('gpurun_out/b.out').read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], d['config'].get('loss'))"
done
