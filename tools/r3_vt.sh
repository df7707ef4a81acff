#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_train.py tests/test_gpu_gemm.py tests/test_gpu_ddp.py -x -q 2>&1 | tail -5 > gpurun_out/r3/vt_tests.log
python bench.py --workload vis_train --no-secondary --no-cpu-baseline > gpurun_out/r3/bench_vt_group.json 2> gpurun_out/r3/bench_vt_group.err
SQ_BWD_NO_GROUP=1 python bench.py --workload vis_train --no-secondary --no-cpu-baseline > gpurun_out/r3/bench_vt_nogroup.json 2> gpurun_out/r3/bench_vt_nogroup.err
cat gpurun_out/r3/vt_tests.log
python -c "
import json
for t in ('group','nogroup'):
    d=json.load(open('gpurun_out/r3/bench_vt_%s.json'%t)); print(t, d['value'], d['ms_per_step'])
"
