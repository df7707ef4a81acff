cd $GRAFT_REPO_ROOT
for v in 448 150; do
SQ_GEMM_RING_MIN_TILES=$v SQ_BENCH_KERNELS=gpurun_out/r2_vt_k$v.json timeout 600 python bench.py --workload vis_train --no-secondary --no-cpu-baseline | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MIN_TILES=$v', d['value'], d['ms_per_step'])"
python -c "
import json; d=json.load(open('gpurun_out/r2_vt_k$v.json'))
for r in d[:4]:
    avg=r['total_ms']/r['count']*1e3
    print(f\"  {r['name'][:44]:44s} n={r['count']:5d} avg={avg:8.1f}us {r['flops']/avg/1e6:7.1f}TF\")"
done
