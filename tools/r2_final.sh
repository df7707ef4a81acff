cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
SECONDS=0
timeout 1500 python bench.py > gpurun_out/r2_bench_final.log 2>&1; echo "bench took $SECONDS s"; tail -1 gpurun_out/r2_bench_final.log > gpurun_out/r2_bench_final.json
python -c "
import json; d=json.load(open('gpurun_out/r2_bench_final.json')); print('value',d['value'],'steps',d['steps'],'region',d['timed_region_s']); print({k:(v.get('value'),v.get('ms_per_step'),v.get('error')) for k,v in d['secondary'].items()}); print(d['roofline']['kernel'],d['roofline']['bound'],d['roofline']['frac'],d['roofline']['end_to_end']); print(d['secondary']['spatial_50k_tiles']['roofline']['end_to_end']); print(d['cpu_baseline']['value'],d['cpu_baseline']['batched_value'])"
