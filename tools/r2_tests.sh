cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r2_tests_full.log 2>&1
echo "rc=$?"
grep -nE "passed|failed|error" gpurun_out/r2_tests_full.log | tail -5
tail -5 gpurun_out/r2_tests_full.log
