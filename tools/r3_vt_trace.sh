#!/bin/bash
# kernel timeline of ONE vis_train step (config 2): rocprofv3 --kernel-trace, last step's kernels in start order
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/vt_trace; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -- python $R/bench.py --workload vis_train --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $O/run.log 2>&1
cd $R
python - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/r3/vt_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find adamw launches: step boundaries
idx = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]          # one full timed step (between two adamw kernels)
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp'])
span = (int(step[-1]['End_Timestamp']) - t0) / 1e3
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3
print(f"step: {len(step)} kernels, span {span:.1f} us, sum of kernel durations {busy:.1f} us")
out = open('gpurun_out/r3/vt_timeline.txt', 'w')
for r in step:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r['Kernel_Name'])[:60]
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    out.write(f"{s:9.1f} {e:9.1f} {e - s:7.1f} q{r.get('Queue_Id','?')} grid={r.get('Grid_Size','?'):>9s} {n}\n")
out.close()
PY
find $O -name "*.csv" -size +2M -delete
