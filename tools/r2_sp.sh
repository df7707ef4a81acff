cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/sp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sp -- python $R/bench.py --workload spatial --no-secondary --no-cpu-baseline --steps 1 --warmup 1 > $R/gpurun_out/sp.log 2>&1
cd $R
f=$(ls gpurun_out/sp/*/*kernel_stats.csv | head -1)
python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:80]:80s} calls={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:8.1f}us {r['Percentage']}%")
PY
find gpurun_out/sp -name "*kernel_trace.csv" -delete
