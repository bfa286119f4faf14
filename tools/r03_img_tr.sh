set -o pipefail
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_round2_gpu.py tests/test_dropin_gpu.py -m gpu -x -q > gpurun_out/r03aa_test.log 2>&1; rc=$?; tail -3 gpurun_out/r03aa_test.log
[ $rc -ne 0 ] && exit $rc
BD_LIB=/root/repo/big_dreamer_amd/libbd_stamps.so timeout -k 10 300 python tools/stamps.py 2>/dev/null | head -32
for i in 1 2 3; do timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r03aa_$i.json 2> gpurun_out/r03aa_$i.err || exit 1
python - <<PY
import json
j = json.loads(open("gpurun_out/r03aa_$i.json").read().strip().splitlines()[-1]); k = j["kernel_ms"]
print(round(j["ms_per_step"], 3), "img_fwd", k["imagine_fwd"], "img_bwd", k["imagine_bwd"], "obs", k["observe_fwd"], k["observe_bwd"])
PY
done
