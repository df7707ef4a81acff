cd $GRAFT_REPO_ROOT
for t in 64 128 256; do echo "seed threads $t"; SQ_KM_SEED_THREADS=$t timeout 300 python tools/kmeans_time.py; done
SQ_KM_SEED_THREADS=64 timeout 900 python -m pytest tests/test_gpu_kmeans.py -x -q 2>&1 | tail -3
