# one-hot weight gradients as segmented sums: unit + Categorical parity tests, then A/B of configs[4] (state, pixel) on one box
set -o pipefail
T=${1:-r03u}
timeout -k 10 900 python -m pytest tests/test_categorical_gpu.py tests/test_categorical_scan_gpu.py -x -q > gpurun_out/${T}_cattest.log 2>&1; rc=$?; tail -4 gpurun_out/${T}_cattest.log
[ $rc -ne 0 ] && exit $rc
for m in state pixel; do
for v in 1 0; do
BD_WGRAD_ONEHOT=$v timeout -k 10 400 python bench.py --categorical $m --steps 12 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_${m}_$v.json 2> gpurun_out/${T}_${m}_$v.err || { tail -5 gpurun_out/${T}_${m}_$v.err; exit 1; }
python - <<PY
import json
j = json.loads(open("gpurun_out/${T}_${m}_$v.json").read().strip().splitlines()[-1])
k = j["kernel_ms"]
print("$m onehot=$v", round(j["ms_per_step"], 3), {x: k[x] for x in ("wgrad_gemm_critic", "wgrad_critic", "wgrad_gemm_actor", "wgrad_actor", "imagine_fwd", "imagine_bwd", "img_heads_bwd") if x in k})
PY
done
done
