set -o pipefail
timeout -k 10 900 python -m pytest tests/test_categorical_scan_gpu.py tests/test_dp_gpu.py -x -q > gpurun_out/r03b_cat_tests.log 2>&1; rc=$?; tail -25 gpurun_out/r03b_cat_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --categorical state --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03b_cat_state.json 2> gpurun_out/r03b_cat_state.err || exit 1
BD_OBS_CLUSTER=0 timeout -k 10 300 python bench.py --categorical state --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03b_cat_state_nocluster.json 2> gpurun_out/r03b_cat_state_nocluster.err || exit 1
python - <<'PY'
import json
for f in ("r03b_cat_state", "r03b_cat_state_nocluster"):
    j = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, round(j["ms_per_step"], 3), {k: v for k, v in j["kernel_ms"].items() if "observe" in k})
PY
