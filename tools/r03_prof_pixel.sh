set -o pipefail
T=${1:-r03h}
R=$(pwd)
OUT=$R/gpurun_out/$T
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pixel_stats -- python3 $R/bench.py --pixel --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pixel_stats.log 2>&1 || { tail -5 $OUT/pixel_stats.log; exit 1; }
cd $R
f=$(ls $OUT/pixel_stats/*/*kernel_stats.csv | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per step (12 steps):", round(tot/12e6,3), "ms")
for r in rows[:26]:
    print(f"{r['Name'][:84]:84s} {int(r['Calls'])/12:5.1f}/step avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.2f}%")
PY
