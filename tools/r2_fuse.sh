cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_resnet.py -x -q 2>&1 | tail -4
for c in 0 1; do
SQ_RESNET_NO_CHAIN=$c SQ_RESNET_STREAMS=1 SQ_BENCH_KERNELS=gpurun_out/r2_kern_chain$c.json timeout 300 python bench.py --steps 2 --warmup 1 --slides 2 --no-cpu-baseline --no-stream --no-secondary > gpurun_out/r2_chain$c.log 2>&1
tail -1 gpurun_out/r2_chain$c.log | cut -c60-110
SQ_RESNET_NO_CHAIN=$c timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r2_chain${c}_full.log 2>&1
tail -1 gpurun_out/r2_chain${c}_full.log | cut -c60-110
done
