#!/bin/bash
# sweep of the ResNet launch-group size (patches per group) for the f16x3 pipeline: does a working set that fits the
# 256 MB memory-side cache beat the larger group's fuller launches?
mkdir -p gpurun_out/r3
for sb in 500 250 200 125 100 50; do
  for ns in 2 1; do
    echo "== sub_batch $sb streams $ns"
    SQ_RESNET_STREAMS=$ns python bench.py --resident --sub-batch $sb --steps 6 --no-accuracy --no-cpu-baseline --no-secondary 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])
"
  done
done
