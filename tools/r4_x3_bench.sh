set -o pipefail
mkdir -p gpurun_out/r4b
python -m pytest tests/test_gpu_bf16x3.py -q -k "first_layer" 2>&1 | tail -3
python bench.py --dtype bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-parity --layers > gpurun_out/r4b/x3.json 2> gpurun_out/r4b/x3_layers.txt || tail -20 gpurun_out/r4b/x3_layers.txt
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-parity > gpurun_out/r4b/bf16.json 2> gpurun_out/r4b/bf16.err
python - <<'PY'
import json
for n in ("x3","bf16"):
    try:
        d=json.load(open("gpurun_out/r4b/%s.json"%n)); print(n, d["ms_per_step"], d.get("roofline",{}).get("achieved"))
    except Exception as e: print(n, "ERR", e)
PY
grep -v "^DICE\|amdgpu" gpurun_out/r4b/x3_layers.txt | head -40
