#!/bin/bash
# x3 (split-fp16) mode: parity tests, then the per-kernel breakdown of the pipeline step (ping-pong vs lockstep schedule)
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_x3.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -8 > gpurun_out/r3/x3_tests.log
SQ_BENCH_KERNELS=gpurun_out/r3/kern_f16x3_pp.json python bench.py --dtype f16x3 --resident --no-secondary --no-cpu-baseline --no-accuracy > gpurun_out/r3/bench_f16x3_pp.json 2> gpurun_out/r3/bench_f16x3_pp.err
if [ -n "$AB" ]; then
SQ_X3_LOCKSTEP=1 SQ_BENCH_KERNELS=gpurun_out/r3/kern_f16x3_ls.json python bench.py --dtype f16x3 --resident --no-secondary --no-cpu-baseline --no-accuracy > gpurun_out/r3/bench_f16x3_ls.json 2> gpurun_out/r3/bench_f16x3_ls.err
fi
cat gpurun_out/r3/x3_tests.log
