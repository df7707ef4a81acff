run() { env $1 python bench.py --no-secondary --no-cpu-baseline ${@:2} 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], d["ms_per_step"], "ms/step")'; }
for a in SQ_SPATIAL_STREAMS=2 SQ_SPATIAL_STREAMS=1 SQ_SPATIAL_STREAMS=3; do echo "[$a] $(run $a --workload spatial)"; done
for bw in 512 2048; do echo "[batch-windows $bw] $(run X=1 --workload spatial --batch-windows $bw)"; done
for sb in 256 500 1000; do echo "[uni sub-batch $sb] $(run X=1 --workload pipeline --embedder uni --slides 2 --uni-sub-batch $sb)"; done
