#!/bin/bash
# Two processes on one GPU, each confined to one half of the CUs (HSA_CU_MASK), each running the f16x3 pipeline: is the sum of
# their rates above the one-process rate?  Control: two processes without masks.  Body of `gpurun -- 'bash tools/cu_split_probe.sh'`.
cd $GRAFT_REPO_ROOT
B="python bench.py --no-secondary --no-cpu-baseline --no-accuracy --slides 8 --steps 6 --warmup 2"
echo "one process, whole chip:"; $B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1
for lohi in "0-127 128-255" "even odd"; do
  set -- $lohi
  if [ $1 = even ]; then
    A=$(python -c "print('0:' + ','.join(str(i) for i in range(0, 256, 2)))"); C=$(python -c "print('0:' + ','.join(str(i) for i in range(1, 256, 2)))")
  else A="0:$1"; C="0:$2"; fi
  echo "two processes, masks $1 / $2:"
  HSA_CU_MASK=$A $B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 > /tmp/pa.txt &
  HSA_CU_MASK=$C $B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 > /tmp/pb.txt &
  wait; cat /tmp/pa.txt /tmp/pb.txt
done
echo "two processes, no masks:"
$B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 > /tmp/pa.txt &
$B 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 > /tmp/pb.txt &
wait; cat /tmp/pa.txt /tmp/pb.txt
