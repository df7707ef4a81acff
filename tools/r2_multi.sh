cd $GRAFT_REPO_ROOT
# two ranks sharing the one GPU (gloo): exercises the self-spawn, rendezvous, barriers, max-over-ranks timing
SQ_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --slides 2 2>&1 | tail -3 | cut -c1-400
SQ_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --workload train_kfold --steps 1 --warmup 0 --epochs 2 2>&1 | tail -3 | cut -c1-500
SQ_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --workload vis_train --steps 10 --warmup 2 2>&1 | tail -2 | cut -c1-300
timeout 600 python bench.py --workload train_kfold --no-cpu-baseline 2>&1 | tail -1 | cut -c1-500
# a launcher-provided world that disagrees with --gpus must be refused
WORLD_SIZE=2 RANK=0 LOCAL_RANK=0 timeout 60 python bench.py --gpus 1 2>&1 | tail -1
