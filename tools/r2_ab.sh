cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_pipeline.py -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for v in base new; do
  cp ab/$v.so sequoia-pub_amd/libsequoia_hip.so
  SQ_BENCH_KERNELS=gpurun_out/r2_ab_$v.json timeout 600 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r2_ab_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r2_ab_$v.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")"
  python -c "
import json; d=json.load(open('gpurun_out/r2_ab_$v.json'))
print('  '+' '.join(f\"{r['name'].replace('btl_','').replace('_P1568000','').replace('_P392000','').replace('_P98000','')}={r['total_ms']/r['count']*1e3:.0f}\" for r in d if 'chain' in r['name'] or 'tail' in r['name']))"
done; done
