#!/usr/bin/env python
"""Print a bench line's value and the per-kernel table written by SQ_BENCH_KERNELS (bench.py)."""
import json, sys
bench, kern = sys.argv[1], sys.argv[2]
slides = float(sys.argv[3]) if len(sys.argv) > 3 else 24.0
try:
    d = json.load(open(bench))
    print("value", d["value"], "ms/step", d["ms_per_step"], "steps", d["steps"])
except Exception as e:
    print("bench line unreadable:", e); print(open(bench.replace(".json", ".err")).read()[-2000:])
k = json.load(open(kern))
tot = sum(r["total_ms"] for r in k)
print(f"instrumented total {tot:.1f} ms = {tot / slides:.2f} ms per slide")
for r in k[:int(sys.argv[4]) if len(sys.argv) > 4 else 28]:
    avg = r["total_ms"] / r["count"]
    print(f'{r["name"]:44s} n={r["count"]:4d} avg_us={avg * 1e3:8.1f} ms/slide={r["total_ms"] / slides:6.2f} share={r["total_ms"] / tot:.3f} TF={r["flops"] / avg / 1e9:7.1f} GB/s={r["bytes"] / avg / 1e6:7.1f}')
