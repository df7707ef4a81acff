cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_ddp.py -q -m gpu 2>&1 | tail -3
for v in 0 1; do
SQ_NO_BUCKETS=$v timeout 600 python bench.py --workload vis_train --no-secondary --no-cpu-baseline | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NO_BUCKETS=$v', d['value'], d['ms_per_step'])"
done
