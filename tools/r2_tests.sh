cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
timeout 900 python bench.py > gpurun_out/r2_bench_default.log 2>&1; tail -1 gpurun_out/r2_bench_default.log
