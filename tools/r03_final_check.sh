timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03ab_gputest.log 2>&1; rc=$?; tail -3 gpurun_out/r03ab_gputest.log
[ $rc -ne 0 ] && exit $rc
for v in 3 2 3 2; do
BD_HOST_AHEAD=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('main host_ahead=$v (20 steps):', round(j['ms_per_step'],3))"
done
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default bench:', round(j['ms_per_step'],3), 'surface', j.get('surface_ms_per_step'), 'secondary', (j.get('secondary') or {}).get('ms_per_step'))"
