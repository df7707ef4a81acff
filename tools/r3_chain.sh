#!/bin/bash
# split-mode chain kernel: bit-identity test, then its launches in the pipeline step
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_x3.py -q -k "chain" 2>&1 | grep -E "passed|failed" | tail -2
SQ_BENCH_KERNELS=gpurun_out/r3/kern_chain.json python bench.py --dtype f16x3 --resident --no-secondary --no-cpu-baseline --no-accuracy > gpurun_out/r3/bench_chain.json 2>/dev/null
python tools/kern_summary.py gpurun_out/r3/bench_chain.json gpurun_out/r3/kern_chain.json 24 40 | grep -E "value|chain|tail|N256_K64|N64_K256|N128_K256"
