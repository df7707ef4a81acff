cd $GRAFT_REPO_ROOT
run() { # name, env..., -- args
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary $EXTRA > gpurun_out/r2_cfg_$name.log 2>&1
  echo "$name: $(tail -1 gpurun_out/r2_cfg_$name.log | cut -c1-120)"
}
EXTRA="" run base SQ_GEMM_RING=0
EXTRA="" run ringk2048 SQ_GEMM_RING_MIN_K=2048
EXTRA="--sub-batch 1000" run sb1000_noring SQ_GEMM_RING=0
EXTRA="--sub-batch 1000" run sb1000_ring2048 SQ_GEMM_RING_MIN_K=2048
EXTRA="--sub-batch 1000" run sb1000_256 SQ_GEMM_RING=0 SQ_GEMM256=1
EXTRA="--sub-batch 1000" run sb1000_256_t256 SQ_GEMM_RING=0 SQ_GEMM256=1 SQ_GEMM256_MIN_TILES=256
EXTRA="--sub-batch 334" run sb334 SQ_GEMM_RING=0
