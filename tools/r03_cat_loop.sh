# Categorical imagination scans on whole rounds of workgroups (tile loop): parity, then A/B of configs[4] (pixel and state) on one box
set -o pipefail
T=${1:-r03t}
timeout -k 10 900 python -m pytest tests/test_categorical_scan_gpu.py -x -q > gpurun_out/${T}_cattest.log 2>&1; rc=$?; tail -4 gpurun_out/${T}_cattest.log
[ $rc -ne 0 ] && exit $rc
for m in state pixel; do
for v in 1 0; do
BD_CAT_TILE_LOOP=$v timeout -k 10 400 python bench.py --categorical $m --steps 12 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_${m}_$v.json 2> gpurun_out/${T}_${m}_$v.err || { tail -5 gpurun_out/${T}_${m}_$v.err; exit 1; }
python - <<PY
import json
j = json.loads(open("gpurun_out/${T}_${m}_$v.json").read().strip().splitlines()[-1])
k = j["kernel_ms"]
print("$m loop=$v", round(j["ms_per_step"], 3), {x: k[x] for x in ("imagine_fwd", "imagine_bwd", "wm_heads_fwd", "observe_fwd", "observe_bwd", "wgrad_critic", "critic_fwd_bwd") if x in k})
PY
done
done
