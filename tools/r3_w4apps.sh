#!/bin/bash
mkdir -p gpurun_out/r3
run() { python bench.py --no-secondary --no-cpu-baseline --no-accuracy "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'])"; }
for w4 in 1 0; do
  echo "W4=$w4 spatial: $(SQ_GEMM_W4=$w4 run --workload spatial)"
  echo "W4=$w4 uni: $(SQ_GEMM_W4=$w4 run --workload pipeline --embedder uni --slides 2 --dtype bf16)"
  echo "W4=$w4 pipeline bf16: $(SQ_GEMM_W4=$w4 run --workload pipeline --dtype bf16 --resident)"
done
echo "GEMM256 spatial: $(SQ_GEMM_W4=0 SQ_GEMM256=1 run --workload spatial)"
echo "GEMM256 uni: $(SQ_GEMM_W4=0 SQ_GEMM256=1 run --workload pipeline --embedder uni --slides 2 --dtype bf16)"
