# one-rank RCCL rehearsal of the data-parallel schedule (every collective issued): one communicator per optimiser (default)
# against one for all, in the bench and from a library user's import order; then the GPU data-parallel tests
for g in 1 0; do
env BD_PHASE_GROUPS=$g BD_FORCE_DP=1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench rehearsal, phase groups $g:', round(j['ms_per_step'],3))"
echo -n "library rehearsal, phase groups $g: "; BD_PHASE_GROUPS=$g RCCL_HIGH=0 timeout -k 10 200 python tools/dp_rehearsal.py 2>&1 | tail -1
done
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench, no collectives:', round(j['ms_per_step'],3))"
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py tests/test_hip_parity.py -m gpu -x -q -k "two_ranks or dp or defer or pipelined" 2>&1 | tail -2
