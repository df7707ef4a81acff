cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_cli.py -q -m gpu 2>&1 | tail -5
for v in 0 1; do
  SQ_RESNET_NO_CHAIN256=$v timeout 600 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r2_c256_$v.log 2>&1
  echo "NO_CHAIN256=$v"; tail -1 gpurun_out/r2_c256_$v.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel'], d['roofline']['avg_us'], d['roofline']['frac'])"
  grep -E "btl_chain|M98000_N1024|M98000_N256_K1024" gpurun_out/r2_c256_$v.log | head -8
done
