#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_x3.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -8 > gpurun_out/r3/x3c_tests.log
SQ_BENCH_KERNELS=gpurun_out/r3/kern_f16x3_halo.json python bench.py --dtype f16x3 --resident --no-secondary --no-cpu-baseline --no-accuracy > gpurun_out/r3/bench_f16x3_halo.json 2> gpurun_out/r3/bench_f16x3_halo.err
cat gpurun_out/r3/x3c_tests.log
