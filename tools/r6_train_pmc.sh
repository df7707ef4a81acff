# PMC passes of the ViS training step (config 2) for profiles/r06_vis_train_bf16_pmc.json (what bench.py's vis_train line reads as roofline.traffic)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06; mkdir -p $O $R/gpurun_out/profiles_r06
TRAIN1="python $R/bench.py --workload vis_train --no-secondary --no-cpu-baseline --steps 3 --warmup 1"
rm -rf $O/pmc_train_*
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_train_$n -- $TRAIN1 > $O/pmc_train_$n.log 2>&1
done
cd $R
python tools/pmc_summary.py r06_vis_train_bf16 gpurun_out/profiles_r06/r06_vis_train_bf16_pmc.json $O/pmc_train_FETCH_SIZE $O/pmc_train_WRITE_SIZE $O/pmc_train_SQ_VALU_MFMA_BUSY_CYCLES \
  "gemm_bf16_M6400_N1024_K1024_b1=gemm_nt_kernel<unsigned short, 2, 2, false:102400" \
  "gemmtn_bf16_M1024_N1024_K6400_b4=gemm_tn_ring_kernel:131072"
cat gpurun_out/profiles_r06/r06_vis_train_bf16_pmc.json | head -40
find $O -name "*.csv" -size +5M -delete
