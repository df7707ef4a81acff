cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r02
bash tools/r2_profile.sh > gpurun_out/r2_profile2.log 2>&1
tail -3 gpurun_out/r2_profile2.log
cd $GRAFT_REPO_ROOT
SQ_BENCH_KERNELS=gpurun_out/r2_kernels_pipeline.json timeout 1500 python bench.py > gpurun_out/r2_bench_final.log 2>&1; tail -1 gpurun_out/r2_bench_final.log > gpurun_out/r2_bench_final.json
SQ_BENCH_KERNELS=gpurun_out/r2_kernels_train.json timeout 600 python bench.py --workload vis_train --no-secondary --no-cpu-baseline > gpurun_out/r2_bench_train.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r2_bench_final.json')); print('value',d['value'],'steps',d['steps'],'region',d['timed_region_s']); print({k:(v.get('value'),v.get('ms_per_step'),v.get('error')) for k,v in d['secondary'].items()}); print(d['roofline']['kernel'],d['roofline']['bound'],d['roofline']['frac'],d['roofline']['avg_us'])"
