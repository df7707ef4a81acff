cd $GRAFT_REPO_ROOT
for mt in 448 256; do
SQ_CONV_HALO_MIN_TILES=$mt timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r2_halo_mt$mt.log 2>&1
echo "min_tiles $mt: $(tail -1 gpurun_out/r2_halo_mt$mt.log | cut -c60-110)"
done
