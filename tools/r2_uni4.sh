cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/unip
SEQUOIA_ALLOW_RANDOM_UNI=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/unip -- python $R/tools/uni_time.py > $R/gpurun_out/unip.log 2>&1
cd $R
tail -3 gpurun_out/unip.log
f=$(ls gpurun_out/unip/*/*kernel_stats.csv | head -1)
python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:70]:70s} calls={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:8.1f}us {r['Percentage']}%")
PY
find gpurun_out/unip -name "*kernel_trace.csv" -delete
