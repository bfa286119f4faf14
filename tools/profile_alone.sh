#!/bin/bash
# Run ON THE GPU BOX from the repo root: rocprofv3 kernel statistics of the dense-chain and weight-gradient kernels ALONE on
# the GPU (tools/tall_probe.py, tools/wgrad_stamps.py) -- the un-contended per-kernel rates behind DESIGN.md section 3.
#   tools/profile_alone.sh r02c   ->  gpurun_out/<tag>/alone_{chain,wgrad}/..._kernel_stats.csv
set -o pipefail
TAG=${1:-r02}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/alone_chain -- python3 $R/tools/tall_probe.py > $OUT/alone_chain.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/alone_wgrad -- python3 $R/tools/wgrad_stamps.py > $OUT/alone_wgrad.log 2>&1 || exit 1
cd $R
echo ALONE_PROFILE_OK
