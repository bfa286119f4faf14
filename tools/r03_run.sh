# usage: bash tools/r03_run.sh <tag> [pytest args...]   -- GPU tests into gpurun_out/<tag>_gputest.log, then the three bench lines
set -o pipefail
T=$1; shift
python -m pytest "$@" -x -q > gpurun_out/${T}_gputest.log 2>&1; rc=$?; tail -6 gpurun_out/${T}_gputest.log
[ $rc -ne 0 ] && exit $rc
bash tools/r03_bench_all.sh $T
