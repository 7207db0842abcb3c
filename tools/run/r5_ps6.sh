set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_round5.py -k plane_serial -x -q 2>&1 | tail -2
for v in "X=1" "X=1"; do
  echo "== x3 $v"; env "$v" timeout -k 10 200 python bench.py --dtype bf16x3 --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --layers > gpurun_out/x3.out 2> gpurun_out/x3.err; grep -E "48->16 @92x92x92 zm" gpurun_out/x3.err | grep -v wgrad | cut -c1-100; python -c "import json; d=json.loads(open('gpurun_out/x3.out').read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], d['config'].get('loss'))"
done
