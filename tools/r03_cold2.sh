for cfg in "--pixel" "--categorical state" "--categorical pixel"; do
for v in 3 0 2; do
BD_HOST_AHEAD=$v timeout -k 10 300 python bench.py $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg host_ahead=$v:', round(j['ms_per_step'],3))"
done
done
