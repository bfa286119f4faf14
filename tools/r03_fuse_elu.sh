# ELU backward fused into the dgrad epilogues: conv + pixel parity tests, then A/B of the pixel step on one box
set -o pipefail
T=${1:-r03s}
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_hip_parity.py tests/test_categorical_scan_gpu.py -x -q -k "conv or pixel or config3 or wgrad or gemm or elu" > gpurun_out/${T}_pixtest.log 2>&1; rc=$?; tail -5 gpurun_out/${T}_pixtest.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
for v in 1 0; do
BD_CONV_FUSE_ELU=$v timeout -k 10 300 python bench.py --pixel --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_pixel_$v$i.json 2> gpurun_out/${T}_pixel_$v$i.err || { tail -5 gpurun_out/${T}_pixel_$v$i.err; exit 1; }
python - <<PY
import json
j = json.loads(open("gpurun_out/${T}_pixel_$v$i.json").read().strip().splitlines()[-1])
print("fuse=$v", round(j["ms_per_step"], 3), {k: v for k, v in j["kernel_ms"].items() if "enc" in k or "dec" in k or "heads_fwd" in k})
PY
done
done
