# FETCH/WRITE bytes of every kernel of a 2-slide pipeline run (separate passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcq; rm -rf $O; mkdir -p $O
PIPE1="python $R/bench.py --no-secondary --no-cpu-baseline --slides 2 --steps 1 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- $PIPE1 > $O/$c.log 2>&1
done
cd $R
python tools/pmc_summary.py pmcq $O/summary.json $O/FETCH_SIZE $O/WRITE_SIZE
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmcq/summary.json'))
for k,v in list(d['kernels'].items())[:28]:
    print(f"{k[:70]:70s} n={v['dispatches']:4d} fetch={v.get('fetch_bytes_avg',0)/1e6:8.1f}MB write={v.get('write_bytes_avg',0)/1e6:8.1f}MB")
PY
find $O -name "*.csv" -size +5M -delete
