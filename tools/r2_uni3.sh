cd $GRAFT_REPO_ROOT
for cfg in "SQ_GEMM256=0 SQ_GEMM_RING=1" "SQ_GEMM256=1 SQ_GEMM_RING=0" "SQ_GEMM256=1 SQ_GEMM_RING=1"; do
env $cfg SQ_BENCH_KERNELS=gpurun_out/r2_uni_k.json timeout 900 python bench.py --embedder uni --slides 2 --no-secondary --no-cpu-baseline > gpurun_out/r2_uni3.log 2>&1
echo "$cfg: $(tail -1 gpurun_out/r2_uni3.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
python -c "
import json; d=json.load(open('gpurun_out/r2_uni_k.json'))
print('   '+' '.join(f\"{r['name'].replace('gemm_bf16_M50432_','').replace('_b1','')}={r['flops']/(r['total_ms']/r['count']*1e3)/1e6:.0f}TF\" for r in d[:5] if 'M50432' in r['name']))"
done
