# K-split observe scan: parity of all cluster forms, phase stamps, A/B of the hand-off forms (one box)
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_round2_gpu.py -m gpu -x -q > gpurun_out/r03r_test.log 2>&1; rc=$?; tail -5 gpurun_out/r03r_test.log
[ $rc -ne 0 ] && exit $rc
BD_LIB=/root/repo/big_dreamer_amd/libbd_stamps.so timeout -k 10 200 python tools/ks_stamps.py > gpurun_out/r03r_ks_stamps.log 2>&1 || { tail -5 gpurun_out/r03r_ks_stamps.log; exit 1; }
cat gpurun_out/r03r_ks_stamps.log
bash tools/r03_ab_main.sh r03r "BD_OBS_KSPLIT=1" "BD_OBS_KSPLIT=2"
