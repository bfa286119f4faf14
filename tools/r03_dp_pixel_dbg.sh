for q in 12 24 32; do
echo -n "library rehearsal, $q queues: "; env GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python tools/dp_rehearsal.py 2>&1 | tail -1
for cfg in "" "--pixel"; do
env GPU_MAX_HW_QUEUES=$q BD_FORCE_DP=1 timeout -k 10 300 python bench.py $cfg --steps 20 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench [$cfg] rehearsal, $q queues:', round(j['ms_per_step'],3))"
done
env GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench, no collectives, $q queues:', round(j['ms_per_step'],3))"
done
