#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/x3pmc; rm -rf $O; mkdir -p $O
T="python $R/tools/x3_pmc_target.py"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/a -- $T > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $O/b -- $T > $O/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES --output-format csv -d $O/c -- $T > $O/c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr --output-format csv -d $O/d -- $T > $O/d.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- $T > $O/s.log 2>&1
cd $R
python - <<'PY'
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r3/x3pmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        n=re.sub(r"\(.*$","",r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ',''))
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    if 'x3' not in k: continue
    c={n:sum(v)/len(v) for n,v in acc[k].items()}
    wc=c.get('SQ_WAVE_CYCLES',1)
    print(k)
    print('   frac of WAVE_CYCLES:', ' '.join(f"{n.replace('SQ_','')}={v/wc:.3f}" for n,v in sorted(c.items()) if n.startswith('SQ_') and n!='SQ_WAVE_CYCLES' and not n.startswith('SQ_INSTS') and 'MFMA' not in n and n!='SQ_WAVES'))
    print('   raw:', ' '.join(f"{n}={v:.4g}" for n,v in sorted(c.items()) if not (n.startswith('SQ_') and n!='SQ_WAVE_CYCLES' and not n.startswith('SQ_INSTS') and 'MFMA' not in n and n!='SQ_WAVES')))
PY
for f in $(find $O/s -name "*kernel_stats.csv"); do head -8 $f; done
find $O -name "*.csv" -size +3M -delete
