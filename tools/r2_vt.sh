cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/vt
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/vt -- python $R/bench.py --workload vis_train --no-secondary --no-cpu-baseline --steps 6 --warmup 3 > $R/gpurun_out/vt.log 2>&1
cd $R
tail -1 gpurun_out/vt.log | cut -c1-200
python - <<'PY'
import csv,glob,collections
f=sorted(glob.glob('gpurun_out/vt/**/*kernel_trace.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
print(rows[0].keys())
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find adamw launches -> step boundaries
ad=[i for i,r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
print('adamw count', len(ad))
i0,i1=ad[-3]+1,ad[-2]+1     # one timed step (before roofline extra steps?)
seg=rows[i0:i1]
t0=int(seg[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in seg)
print('step span us', (t1-t0)/1e3, 'kernels', len(seg))
byq=collections.defaultdict(float)
for r in seg: byq[r.get('Queue_Id','?')]+= (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
print('busy per queue', dict(byq))
# union busy time
ev=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in seg)
busy=0; cs,ce=ev[0]
for s,e in ev[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print('union busy us', busy/1e3, 'idle', (t1-t0-busy)/1e3)
for r in seg:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:44]
    print(f"{(s-t0)/1e3:8.1f} {(e-s)/1e3:7.1f} q{r.get('Queue_Id','?')} {n}")
PY
