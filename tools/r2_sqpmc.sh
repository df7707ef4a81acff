cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sqpmc; rm -rf $O; mkdir -p $O
PIPE1="python $R/bench.py --no-secondary --no-cpu-baseline --slides 2 --steps 1 --warmup 1"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/a -- $PIPE1 > $O/a.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $O/b -- $PIPE1 > $O/b.log 2>&1
cd $R
python - <<'PY'
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/sqpmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        n=re.sub(r"\(.*$","",r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ',''))
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for k in ['btl_tail_kernel<64, false>','btl_tail_kernel<128, false>','btl_chain_kernel<128, 4>','btl_chain256_kernel','conv_halo_kernel','gemm_ring_kernel<0, true>']:
    if k not in acc: print('missing',k); continue
    c={n:sum(v)/len(v) for n,v in acc[k].items()}
    wc=c.get('SQ_WAVE_CYCLES',1)
    print(k)
    print('   ', ' '.join(f"{n.replace('SQ_','')}={v/wc:.3f}" for n,v in sorted(c.items()) if n!='SQ_WAVE_CYCLES' and not n.startswith('SQ_INSTS')))
    print('   ', ' '.join(f"{n.replace('SQ_','')}={v:.3g}" for n,v in sorted(c.items()) if n.startswith('SQ_INSTS') or n=='SQ_WAVE_CYCLES'))
PY
find $O -name "*.csv" -size +3M -delete
