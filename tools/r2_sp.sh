cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_spatial.py tests/test_gpu_vis.py tests/test_abi.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --workload spatial --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
