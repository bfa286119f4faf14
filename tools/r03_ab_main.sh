# A/B of the main config on ONE box: usage bash tools/r03_ab_main.sh <tag> "<ENV_A>" "<ENV_B>"  (each run twice, interleaved)
set -o pipefail
T=$1; A=$2; B=$3
for i in 1 2; do
for v in A B; do
  if [ $v = A ]; then E="$A"; else E="$B"; fi
  env $E timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/${T}_$v$i.json 2> gpurun_out/${T}_$v$i.err || { tail -3 gpurun_out/${T}_$v$i.err; exit 1; }
  python - <<PY
import json
j = json.loads(open("gpurun_out/${T}_$v$i.json").read().strip().splitlines()[-1])
k = j["kernel_ms"]
print("$v$i [$E]", round(j["ms_per_step"], 3), "obs_fwd", k["observe_fwd"], "obs_bwd", k["observe_bwd"], "img_fwd", k["imagine_fwd"], "img_bwd", k["imagine_bwd"])
PY
done
done
