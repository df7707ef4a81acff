#!/bin/bash
# round-3 baseline: GPU tests, then the per-kernel breakdown of the parity (f16x3) and bf16 pipeline steps
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3/tests.log
for dt in f16x3 bf16; do
  SQ_BENCH_KERNELS=gpurun_out/r3/kern_$dt.json python bench.py --dtype $dt --resident --no-secondary --no-cpu-baseline --no-accuracy > gpurun_out/r3/bench_$dt.json 2> gpurun_out/r3/bench_$dt.err
done
SQ_BENCH_KERNELS=gpurun_out/r3/kern_vis_train.json python bench.py --workload vis_train --no-secondary --no-cpu-baseline > gpurun_out/r3/bench_vis_train.json 2> gpurun_out/r3/bench_vis_train.err
cat gpurun_out/r3/tests.log
