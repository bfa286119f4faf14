set -o pipefail
T=${1:-r03c}
timeout -k 10 500 python bench.py --steps 50 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || { tail -5 gpurun_out/${T}_bench.err; exit 1; }
timeout -k 10 300 python bench.py --categorical pixel --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_cat_pixel.json 2> gpurun_out/${T}_cat_pixel.err || { tail -5 gpurun_out/${T}_cat_pixel.err; exit 1; }
timeout -k 10 300 python bench.py --categorical state --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_cat_state.json 2> gpurun_out/${T}_cat_state.err || { tail -5 gpurun_out/${T}_cat_state.err; exit 1; }
python - <<PY
import json
T="${T}"
j = json.loads(open(f"gpurun_out/{T}_bench.json").read().strip().splitlines()[-1])
print("main", round(j["ms_per_step"], 3), "surface", j.get("surface", {}).get("ms_per_step"), j.get("surface", {}).get("ms_per_step_burst50"))
print(" kernel_ms", j["kernel_ms"])
print(" kernel_tflops", j["kernel_tflops"], j.get("kernel_tflops_refused"))
s = j.get("secondary", {})
print("pixel", s.get("ms_per_step"), s.get("kernel_ms"), s.get("roofline", {}).get("frac"))
for f in ("cat_pixel", "cat_state"):
    c = json.loads(open(f"gpurun_out/{T}_{f}.json").read().strip().splitlines()[-1])
    print(f, round(c["ms_per_step"], 3), c["kernel_ms"])
PY
