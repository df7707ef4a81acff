#!/bin/bash
# Power draw and clocks (rocm-smi samples every 0.5 s) while a workload runs: bash tools/power_probe.sh <bench.py arguments...>
cd $GRAFT_REPO_ROOT
python bench.py --no-secondary --no-cpu-baseline --no-accuracy "$@" > /tmp/pp.json 2>/dev/null &
P=$!
sleep 4
for i in $(seq 1 60); do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use" | tr '\n' ' ' | sed 's/  */ /g'; echo
  sleep 0.4
  kill -0 $P 2>/dev/null || break
done
wait $P
grep -o '"value": [0-9.]*' /tmp/pp.json | head -1; grep -o '"ms_per_step": [0-9.]*' /tmp/pp.json | head -1
