cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --embedder uni --slides 2 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-700
timeout 1200 python bench.py 2>&1 | tail -1 > gpurun_out/r2_bench_full.json; python -c "
import json; d=json.load(open('gpurun_out/r2_bench_full.json')); print('value',d['value'],'steps',d['steps'],'region',d['timed_region_s']); print({k:(v.get('value'),v.get('ms_per_step'),v.get('error')) for k,v in d['secondary'].items()}); print(d['roofline']['kernel'],d['roofline']['frac'],d['roofline']['end_to_end']); print(d['cpu_baseline']['value'],d['cpu_baseline']['batched_value'],d['cpu_baseline']['cores'])"
