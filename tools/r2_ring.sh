cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_resnet.py -x -q 2>&1 | tail -5
timeout 600 python tools/gemm_probe.py ring > gpurun_out/r2_ring_probe.log 2>&1; cat gpurun_out/r2_ring_probe.log
for r in 1 0; do
  SQ_GEMM_RING=$r SQ_RESNET_STREAMS=1 SQ_BENCH_KERNELS=gpurun_out/r2_kern_ring$r.json timeout 300 python bench.py --steps 2 --warmup 1 --slides 2 --no-cpu-baseline --no-stream --no-secondary > gpurun_out/r2_ring$r.log 2>&1
  tail -1 gpurun_out/r2_ring$r.log | cut -c1-160
  SQ_GEMM_RING=$r timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r2_ring${r}_full.log 2>&1
  tail -1 gpurun_out/r2_ring${r}_full.log | cut -c1-160
done
