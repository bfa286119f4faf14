# the driver's invocation (--steps 20 --warmup 5) with and without the host throttle, one box; then the default run
for v in 4 0 4 0 3 6; do
BD_HOST_AHEAD=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('host_ahead=$v (20 steps):', round(j['ms_per_step'],3))"
done
for v in 4 0; do
BD_HOST_AHEAD=$v timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('host_ahead=$v (50 steps):', round(j['ms_per_step'],3), 'surface', j.get('surface_ms_per_step'))"
done
