cd $GRAFT_REPO_ROOT
for v in 512 256; do
  SQ_GEMM_RING_MIN_K=$v SQ_BENCH_KERNELS=gpurun_out/r2_mink_k$v.json timeout 600 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r2_mink.log 2>&1
  echo "MIN_K=$v: $(tail -1 gpurun_out/r2_mink.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")"
  python -c "
import json; d=json.load(open('gpurun_out/r2_mink_k$v.json'))
print('  '+' '.join(f\"{r['name'].replace('_b1','')}={r['total_ms']/r['count']*1e3:.0f}\" for r in d if ('K256' in r['name'] or 'K512' in r['name'] or 'K128' in r['name'])))"
done
