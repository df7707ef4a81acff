cd $GRAFT_REPO_ROOT
for v in 0 1; do
echo "SQ_UNI_FP32_STREAM=$v"
SQ_UNI_FP32_STREAM=$v timeout 900 python -m pytest tests/test_gpu_uni.py -q -m gpu -s 2>&1 | grep -i -E "err|passed|failed" | head -8
SQ_UNI_FP32_STREAM=$v SEQUOIA_ALLOW_RANDOM_UNI=1 timeout 600 python tools/uni_time.py 2>&1 | tail -1
done
