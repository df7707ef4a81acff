#!/bin/bash
# SQ / LDS counters of the weight-gradient (TN) kernels on the grouped 4 x (1024 x 1024 x 6400) launch: body of
# `gpurun -- 'bash tools/r4_tnpmc.sh'` (separate --pmc passes with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4/tnpmc; rm -rf $O; mkdir -p $O
T="python $R/tools/tn_probe.py pmc"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/a -- $T > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $O/b -- $T > $O/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/c -- $T > $O/c.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/r4_tn_sq_counters.txt
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r4/tnpmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        n=re.sub(r"\(.*$","",r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ',''))
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    if 'gemm_tn' not in k: continue
    c={n:sum(v)/len(v) for n,v in acc[k].items()}
    wc=c.get('SQ_WAVE_CYCLES',1)
    print(k)
    print('   frac of WAVE_CYCLES:', ' '.join(f"{n.replace('SQ_','')}={v/wc:.3f}" for n,v in sorted(c.items()) if n.startswith('SQ_') and n!='SQ_WAVE_CYCLES' and not n.startswith('SQ_INSTS') and 'MFMA' not in n and n!='SQ_WAVES'))
    print('   raw:', ' '.join(f"{n}={v:.4g}" for n,v in sorted(c.items())))
PY
tail -3 $O/c.log
find $O -name "*.csv" -size +3M -delete
