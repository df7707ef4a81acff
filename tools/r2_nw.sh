cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 4 8 0; do
  SQ_CHAIN_NW=$v SQ_BENCH_KERNELS=gpurun_out/r2_nw_k.json timeout 600 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r2_nw.log 2>&1
  echo "NW=$v: $(tail -1 gpurun_out/r2_nw.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")"
  python -c "
import json; d=json.load(open('gpurun_out/r2_nw_k.json'))
print('  '+' '.join(f\"{r['name'].replace('btl_','').replace('_P1568000','').replace('_P392000','').replace('_P98000','')}={r['total_ms']/r['count']*1e3:.0f}\" for r in d if 'chain_c128' in r['name']))"
done; done
