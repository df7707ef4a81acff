#!/bin/bash
# Per-kernel-class times of the f16x3 pipeline with the whole chip and with half of its CUs (HSA_CU_MASK): do the HBM-bound
# classes lose less than the MFMA-bound ones?  Body of `gpurun -- 'bash tools/cu_mask_probe.sh'`.
cd $GRAFT_REPO_ROOT
for m in full half; do
  if [ $m = half ]; then export HSA_CU_MASK=0:0-127; fi
  SQ_BENCH_KERNELS=gpurun_out/cu_$m.json python bench.py --no-secondary --no-cpu-baseline --no-accuracy --slides 4 --steps 2 --warmup 1 2>/dev/null | grep -o '"value": [0-9.]*' | head -1
done
unset HSA_CU_MASK
python - <<'PY'
import json
a = {r['name']: r for r in json.load(open('gpurun_out/cu_full.json'))}
b = {r['name']: r for r in json.load(open('gpurun_out/cu_half.json'))}
tot = sum(r['total_ms'] for r in a.values())
print(f"{'class':44s} {'share':>6s} {'full us':>9s} {'half us':>9s} {'ratio':>6s}")
for n, r in sorted(a.items(), key=lambda kv: -kv[1]['total_ms'])[:22]:
    if n not in b: continue
    fa, fb = r['total_ms'] / r['count'] * 1e3, b[n]['total_ms'] / b[n]['count'] * 1e3
    print(f"{n:44s} {100 * r['total_ms'] / tot:5.1f}% {fa:9.1f} {fb:9.1f} {fb / fa:6.2f}")
PY
