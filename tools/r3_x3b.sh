#!/bin/bash
# x3: both block shapes through the parity tests, then the pipeline step with the 128-row shape up to K = $1 ...
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_x3.py -x -q 2>&1 | tail -8 > gpurun_out/r3/x3b_tests.log
for mk in "$@"; do
SQ_X3_SMALL_MAXK=$mk SQ_BENCH_KERNELS=gpurun_out/r3/kern_f16x3_mk$mk.json python bench.py --dtype f16x3 --resident --no-secondary --no-cpu-baseline --no-accuracy > gpurun_out/r3/bench_f16x3_mk$mk.json 2> gpurun_out/r3/bench_f16x3_mk$mk.err
done
cat gpurun_out/r3/x3b_tests.log
