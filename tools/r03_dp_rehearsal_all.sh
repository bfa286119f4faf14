# one-rank RCCL rehearsal on every configuration (configs[3] / configs[4] are multi-GPU configurations): collectives issued
# against not issued, one box
for cfg in "" "--pixel" "--categorical state" "--categorical pixel"; do
for f in 1 0; do
env BD_FORCE_DP=$f timeout -k 10 300 python bench.py $cfg --steps 20 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg] collectives issued=$f:', round(j['ms_per_step'],3))"
done
done
