set -o pipefail
mkdir -p gpurun_out/r4b
python bench.py --dtype bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-parity --layers > gpurun_out/r4b/x3.json 2> gpurun_out/r4b/x3_layers.txt && \
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-parity --layers > gpurun_out/r4b/bf16.json 2> gpurun_out/r4b/bf16_layers.txt && \
python - <<'PY'
import json
for n in ("x3","bf16"):
    d=json.load(open("gpurun_out/r4b/%s.json"%n)); print(n, d["ms_per_step"], d.get("roofline",{}).get("achieved"), {k:round(v["time_s_per_step"]*1e3,3) for k,v in d.get("kernels",{}).items()})
PY
