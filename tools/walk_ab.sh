# A/B of the 56x56 tails' tile walk (SQ_X3_TAIL_XCD_WALK: 0 plain, 1 one contiguous run per XCD, n > 1 chunks of n tiles): rate, ms/step,
# counter traffic of the three tail classes and their kernel times
mkdir -p gpurun_out/r5
for w in "$@"; do
  out=$(SQ_X3_TAIL_XCD_WALK=$w SQ_BENCH_KERNELS=gpurun_out/r5/walk_kern_$w.json python bench.py --no-secondary --no-cpu-baseline --no-accuracy 2>/dev/null | tail -1)
  echo "[walk $w] $(echo "$out" | python -c '
import sys,json
d=json.loads(sys.stdin.read())
t=d["roofline"].get("traffic_by_class",{})
print(d["value"], d["ms_per_step"], {k[11:]:(round(v["hbm_bytes"]/1e9,2), v["ratio"]) for k,v in t.items() if k.startswith("tail")})')" $(python -c "
import json
d=json.load(open('gpurun_out/r5/walk_kern_$w.json'))
print([(r['name'][11:], round(r['total_ms']/r['count']*1e3,1)) for r in d if r['name'].startswith('tail')])")
done
