cd $GRAFT_REPO_ROOT
python tools/kmeans_time.py
timeout 600 python bench.py --no-secondary --no-cpu-baseline | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])"
timeout 900 python -m pytest tests/test_gpu_cli.py -q -m gpu 2>&1 | tail -2
