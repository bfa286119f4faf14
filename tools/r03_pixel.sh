set -o pipefail
T=${1:-r03e}
python -m pytest tests/test_conv_gpu.py tests/test_hip_parity.py -x -q -k "conv or pixel or config3 or wgrad or gemm" > gpurun_out/${T}_pixtest.log 2>&1; rc=$?; tail -5 gpurun_out/${T}_pixtest.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
timeout -k 10 300 python bench.py --pixel --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_pixel$i.json 2> gpurun_out/${T}_pixel$i.err || { tail -5 gpurun_out/${T}_pixel$i.err; exit 1; }
python - <<PY
import json
j = json.loads(open("gpurun_out/${T}_pixel$i.json").read().strip().splitlines()[-1])
print("pixel", round(j["ms_per_step"], 3), j["roofline"]["frac"], {k: v for k, v in j["kernel_ms"].items() if "wgrad" in k or "enc" in k or "dec" in k or "heads_fwd" in k})
PY
done
python tools/conv_probe.py 2>/dev/null | sed -n '/weight gradients/,$p' | tee gpurun_out/${T}_conv_probe.log
