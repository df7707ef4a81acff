cd $GRAFT_REPO_ROOT
for v in 0 2; do
  SQ_CHAIN_DBG=$v SQ_BENCH_KERNELS=gpurun_out/r2_dbg_k$v.json timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 6 > gpurun_out/r2_dbg_$v.log 2>&1
  python -c "
import json; d=json.load(open('gpurun_out/r2_dbg_k$v.json'))
print('dbg=$v', ' '.join(f\"{r['name'].replace('btl_chain_','').replace('_P392000','')}={r['total_ms']/r['count']*1e3:.0f}\" for r in d if 'chain_c128' in r['name']))"
done
