set -o pipefail
V=$PWD/stroke-prediction_amd/lib/variants/psnohalf.so
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_bf16x3.py -x -q > gpurun_out/t.log 2>&1; tail -1 gpurun_out/t.log
for v in "SP_LIB_PATH=$V" "X=1" "SP_LIB_PATH=$V" "X=1"; do
  echo "== x3 ${v:0:12}"; env "$v" timeout -k 10 200 python bench.py --dtype bf16x3 --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --layers > gpurun_out/b.out 2> gpurun_out/b.err; grep -E "conv_igemm  48->16" gpurun_out/b.err | cut -c1-100; python -c "import json; d=json.loads(open('gpurun_out/b.out').read().strip().splitlines()[-1]); print('x3 ms', d['ms_per_step'], d['config'].get('loss'))"
done
