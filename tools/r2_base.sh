set -x
cd $GRAFT_REPO_ROOT
for sb in 500 250 125 64; do
  python bench.py --workload pipeline --steps 4 --warmup 1 --no-cpu-baseline --sub-batch $sb > gpurun_out/r2_base_sb$sb.log 2>&1
  tail -1 gpurun_out/r2_base_sb$sb.log | cut -c1-400
done
python bench.py --workload pipeline --steps 4 --warmup 1 --no-cpu-baseline --from-host > gpurun_out/r2_base_fromhost.log 2>&1
tail -1 gpurun_out/r2_base_fromhost.log | cut -c1-300
SQ_RESNET_STREAMS=1 python bench.py --workload pipeline --steps 4 --warmup 1 --no-cpu-baseline --sub-batch 64 > gpurun_out/r2_base_sb64_1s.log 2>&1
tail -1 gpurun_out/r2_base_sb64_1s.log | cut -c1-300
