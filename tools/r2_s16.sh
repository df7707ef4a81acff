cd $GRAFT_REPO_ROOT
for v in 0 1; do
echo "SQ_VIS_FP32_STREAM=$v"
SQ_VIS_FP32_STREAM=$v timeout 900 python -m pytest tests/test_gpu_vis.py tests/test_gpu_spatial.py tests/test_gpu_pipeline.py -q -m gpu -s 2>&1 | grep -i -E "bf16|passed|failed" | head -12
SQ_VIS_FP32_STREAM=$v timeout 900 python bench.py --workload spatial --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('spatial', d['value'], d['ms_per_step'])"
done
