#!/bin/bash
mkdir -p gpurun_out/r3
for cfg in "1 1000" "2 500" "3 334" "4 250" "2 250" "2 1000"; do
  set -- $cfg
  v=$(SQ_RESNET_STREAMS=$1 python bench.py --dtype f16x3 --resident --no-secondary --no-cpu-baseline --no-accuracy --sub-batch $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "streams $1 sub-batch $2: $v"
done
