set -o pipefail
timeout -k 10 400 python -m pytest tests/test_gpu_bf16x3.py -x -q 2>&1 | tail -3
for v in "SP_ZM_PSER=" "SP_HL_PSER_SLICES=0" "X=1" "SP_ZM_PSER=" "X=1"; do
  echo "== x3 $v"; env "$v" timeout -k 10 200 python bench.py --dtype bf16x3 --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --layers > gpurun_out/x3.out 2> gpurun_out/x3.err; grep -E "48->16 @92x92x92 zm|96->32 @50x50x50 " gpurun_out/x3.err | grep -v wgrad | cut -c1-100; python -c "import json; d=json.loads(open('gpurun_out/x3.out').read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], d['config'].get('loss'))" || tail -5 gpurun_out/x3.err
done
