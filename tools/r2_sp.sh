cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_vis.py tests/test_gpu_spatial.py tests/test_gpu_pipeline.py tests/test_gpu_train.py -q -m gpu 2>&1 | tail -2
SQ_BENCH_KERNELS=gpurun_out/r2_sp_k.json timeout 900 python bench.py --workload spatial --no-secondary --no-cpu-baseline > gpurun_out/r2_sp.log 2>&1
tail -1 gpurun_out/r2_sp.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('spatial', d['value'], d['ms_per_step'])"
python -c "
import json; d=json.load(open('gpurun_out/r2_sp_k.json')); tot=sum(r['total_ms'] for r in d)
for r in d[:4]:
    avg=r['total_ms']/r['count']*1e3
    print(f\"{r['name'][:44]:44s} n={r['count']:5d} avg={avg:8.1f}us share={r['total_ms']/tot:.3f} {r['flops']/avg/1e6:7.1f}TF\")"
