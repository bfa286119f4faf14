set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r03a_gputest.log 2>&1; rc=$?; tail -8 gpurun_out/r03a_gputest.log
[ $rc -ne 0 ] && exit $rc
make -C big_dreamer_amd/csrc -j16 stamps > gpurun_out/r03a_stamps_build.log 2>&1 || { tail -20 gpurun_out/r03a_stamps_build.log; exit 1; }
BD_LIB=big_dreamer_amd/libbd_stamps.so timeout -k 10 200 python tools/cat_stamps.py > gpurun_out/r03a_cat_stamps.log 2>&1; tail -60 gpurun_out/r03a_cat_stamps.log
