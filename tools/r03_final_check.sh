# final check of the round, default environment: GPU tests, smoke, the default bench line, one-rank RCCL rehearsal of every
# configuration, the library-order rehearsal, a few CLI steps
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03ad_gputest.log 2>&1; rc=$?; tail -2 gpurun_out/r03ad_gputest.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1
timeout -k 10 300 python bench.py 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default bench:', round(j['ms_per_step'],3), 'ms/step', round(j['value']), j['unit'], '| surface', j.get('surface_ms_per_step'), '| secondary', (j.get('secondary') or {}).get('ms_per_step'), '| roofline', round(j['roofline']['frac'],3), '| cpu', round(j['cpu_baseline']['value']))"
bash tools/r03_dp_rehearsal_all.sh
echo -n "library-order rehearsal: "; timeout -k 10 200 python tools/dp_rehearsal.py 2>&1 | tail -1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver invocation (20 steps, warmup 5):', round(j['ms_per_step'],3))"
