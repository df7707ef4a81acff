cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_resnet.py -x -q 2>&1 | tail -8
for f in 0 1; do
  SQ_RESNET_NO_FUSE=$f SQ_RESNET_STREAMS=1 SQ_BENCH_KERNELS=gpurun_out/r2_kern_fuse$f.json timeout 300 python bench.py --workload pipeline --steps 2 --warmup 1 --slides 2 --no-cpu-baseline --no-stream > gpurun_out/r2_fuse$f.log 2>&1
  tail -1 gpurun_out/r2_fuse$f.log | cut -c1-200
  SQ_RESNET_NO_FUSE=$f timeout 300 python bench.py --workload pipeline --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2_fuse${f}_full.log 2>&1
  tail -1 gpurun_out/r2_fuse${f}_full.log | cut -c1-200
done
