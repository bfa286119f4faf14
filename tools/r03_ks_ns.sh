set -o pipefail
timeout -k 10 700 python -m pytest tests/test_hip_parity.py tests/test_round2_gpu.py -m gpu -x -q > gpurun_out/r03x_test.log 2>&1; rc=$?; tail -4 gpurun_out/r03x_test.log
[ $rc -ne 0 ] && exit $rc
for m in 3 1; do
BD_OBS_KSPLIT=$m BD_LIB=/root/repo/big_dreamer_amd/libbd_stamps.so timeout -k 10 200 python tools/ks_stamps.py > gpurun_out/r03x_ks_stamps_$m.log 2>&1 || { tail -5 gpurun_out/r03x_ks_stamps_$m.log; exit 1; }
echo "== BD_OBS_KSPLIT=$m"; tail -25 gpurun_out/r03x_ks_stamps_$m.log
done
bash tools/r03_ab_main.sh r03x "BD_OBS_KSPLIT=3" "BD_OBS_KSPLIT=1"
