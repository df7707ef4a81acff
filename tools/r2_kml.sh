cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_pipeline.py -q -m gpu 2>&1 | tail -3
python tools/km_lone.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kml
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kml -- python $R/tools/km_lone.py > $R/gpurun_out/kml.log 2>&1
cd $R
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/kml/**/*kernel_trace.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'km_center_kernel' in r['Kernel_Name']]
i0=idx[-1]
t0=int(rows[i0]['Start_Timestamp']); prev_end=t0
for r in rows[i0:]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:40]
    print(f"{(s-t0)/1e3:9.1f}us gap={(s-prev_end)/1e3:7.1f} dur={(e-s)/1e3:7.1f}  {n}")
    prev_end=e
PY
