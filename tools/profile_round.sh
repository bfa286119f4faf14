#!/bin/bash
# Run ON THE GPU BOX from the repo root: bench line + rocprofv3 kernel stats + separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ), all into gpurun_out/<tag>/.  Summarise with tools/summarize_profile.py.
#   tools/profile_round.sh r02a          main config (BASELINE configs[1]) + pixel config (configs[2]) passes
#   tools/profile_round.sh r02a main     main config only;   ... pixel: configs[2] passes only;   ... cat: configs[4] passes only
set -o pipefail
TAG=${1:-r02}
WHAT=${2:-all}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
B="--no-cpu-baseline --no-secondary"
if [ "$WHAT" = "all" ] || [ "$WHAT" = "main" ]; then
timeout -k 10 500 python3 bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 2 $B > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 $B > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 $B > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 $B > $OUT/pmc_sq.log 2>&1 || exit 1
# the same step on ONE stream (BD_PIPELINE=0): every kernel alone on the GPU -- the un-contended durations
BD_PIPELINE=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial_stats -- python3 $R/bench.py --steps 10 --warmup 2 $B > $OUT/serial_stats.log 2>&1 || exit 1
cd $R
echo MAIN_PROFILE_OK
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "pixel" ]; then
cd $R
timeout -k 10 300 python3 bench.py --pixel --steps 20 --warmup 3 --no-secondary > $OUT/pixel_bench.json 2> $OUT/pixel_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pixel_stats -- python3 $R/bench.py --pixel --steps 10 --warmup 2 $B > $OUT/pixel_stats.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pixel_pmc_fetch -- python3 $R/bench.py --pixel --steps 3 --warmup 1 $B > $OUT/pixel_pmc_fetch.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pixel_pmc_write -- python3 $R/bench.py --pixel --steps 3 --warmup 1 $B > $OUT/pixel_pmc_write.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pixel_pmc_sq -- python3 $R/bench.py --pixel --steps 3 --warmup 1 $B > $OUT/pixel_pmc_sq.log 2>&1 || exit 1
cd $R
echo PIXEL_PROFILE_OK
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "cat" ]; then
cd $R
C="--categorical pixel"
timeout -k 10 300 python3 bench.py $C --steps 20 --warmup 3 > $OUT/cat_bench.json 2> $OUT/cat_bench.err || exit 1
timeout -k 10 300 python3 bench.py --categorical state --steps 20 --warmup 3 --no-cpu-baseline > $OUT/cat_state_bench.json 2> $OUT/cat_state_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cat_stats -- python3 $R/bench.py $C --steps 8 --warmup 2 $B > $OUT/cat_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cat_pmc_fetch -- python3 $R/bench.py $C --steps 3 --warmup 1 $B > $OUT/cat_pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/cat_pmc_write -- python3 $R/bench.py $C --steps 3 --warmup 1 $B > $OUT/cat_pmc_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/cat_pmc_sq -- python3 $R/bench.py $C --steps 3 --warmup 1 $B > $OUT/cat_pmc_sq.log 2>&1 || exit 1
# the same latents on state observations: counters only (the kernels differ in their row counts' neighbours, not in code)
C2="--categorical state"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/catstate_pmc_fetch -- python3 $R/bench.py $C2 --steps 3 --warmup 1 $B > $OUT/catstate_pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/catstate_pmc_write -- python3 $R/bench.py $C2 --steps 3 --warmup 1 $B > $OUT/catstate_pmc_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/catstate_pmc_sq -- python3 $R/bench.py $C2 --steps 3 --warmup 1 $B > $OUT/catstate_pmc_sq.log 2>&1 || exit 1
cd $R
echo CAT_PROFILE_OK
fi
echo PROFILE_OK
