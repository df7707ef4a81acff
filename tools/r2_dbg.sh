cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_resnet.py -q -m gpu 2>&1 | tail -2
for v in 0 7; do
  SQ_CHAIN256_DBG=$v SQ_BENCH_KERNELS=gpurun_out/r2_dbg_k$v.json timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 6 > gpurun_out/r2_dbg_$v.log 2>&1
  python -c "
import json; d=json.load(open('gpurun_out/r2_dbg_k$v.json'))
for r in d:
    if 'chain_c256' in r['name']: print('dbg=$v', r['name'], round(r['total_ms']/r['count']*1e3,1))"
  tail -1 gpurun_out/r2_dbg_$v.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"
done
