cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
timeout 300 python tools/kmeans_time.py
