# A/B helper for gpurun: tools/ab.sh "<env assignments A>" "<env assignments B>" <bench.py arguments...>  -> value / ms_per_step of each arm, twice (A B A B)
A="$1"; B="$2"; shift 2
for arm in "$A" "$B" "$A" "$B"; do
  out=$(env $arm python bench.py --no-secondary --no-cpu-baseline --no-measure-traffic "$@" 2>/dev/null | tail -1)
  echo "[$arm] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], d["ms_per_step"], "ms/step")')"
done
