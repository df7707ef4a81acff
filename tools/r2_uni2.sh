cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_uni.py tests/test_gpu_vis.py tests/test_gpu_spatial.py -q -m gpu 2>&1 | tail -3
SQ_BENCH_KERNELS=gpurun_out/r2_uni_k.json timeout 900 python bench.py --embedder uni --slides 2 --no-secondary --no-cpu-baseline > gpurun_out/r2_uni2.log 2>&1
tail -1 gpurun_out/r2_uni2.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('uni', d['value'], d['ms_per_step'])"
python -c "
import json; d=json.load(open('gpurun_out/r2_uni_k.json')); tot=sum(r['total_ms'] for r in d)
for r in d[:6]:
    avg=r['total_ms']/r['count']*1e3
    print(f\"{r['name'][:44]:44s} n={r['count']:5d} avg={avg:8.1f}us share={r['total_ms']/tot:.3f} {r['flops']/avg/1e6:7.1f}TF\")"
timeout 900 python bench.py --workload spatial --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('spatial', d['value'], d['unit'], d['ms_per_step'])"
