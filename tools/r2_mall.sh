cd $GRAFT_REPO_ROOT
for sb in 500 100 50 25; do
  SQ_RESNET_STREAMS=1 SQ_BENCH_KERNELS=gpurun_out/r2_kern_sb$sb.json python bench.py --workload pipeline --steps 2 --warmup 1 --slides 2 --no-cpu-baseline --no-stream --sub-batch $sb > gpurun_out/r2_mall_sb$sb.log 2>&1
  tail -1 gpurun_out/r2_mall_sb$sb.log | cut -c1-200
done
