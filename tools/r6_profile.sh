# rocprofv3 evidence for round 6 (run on the GPU box through gpurun; raw outputs under gpurun_out/prof_r06/, summaries -> gpurun_out/profiles_r06/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06; mkdir -p $O $R/gpurun_out/profiles_r06
PIPE="python $R/bench.py --no-secondary --no-cpu-baseline --no-accuracy --no-measure-traffic --steps 3 --warmup 1"
rm -rf $O/pipe $O/pipe_serial $O/pmc_pipe_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe -- $PIPE > $O/pipe.log 2>&1
SQ_RESNET_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe_serial -- $PIPE --no-stream > $O/pipe_serial.log 2>&1
PIPE1="python $R/bench.py --no-secondary --no-cpu-baseline --no-accuracy --no-measure-traffic --slides 2 --steps 1 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_pipe_$n -- $PIPE1 > $O/pmc_pipe_$n.log 2>&1
done
cd $R
for w in pipe pipe_serial; do
  f=$(ls $O/$w/*/*kernel_stats.csv 2>/dev/null | head -1)
  case $w in pipe) t=r06_pipeline_f16x3_kernel_stats.csv;; pipe_serial) t=r06_pipeline_f16x3_kernel_stats_serial.csv;; esac
  [ -n "$f" ] && cp $f gpurun_out/profiles_r06/$t
done
python tools/pmc_summary.py r06_pipeline_f16x3 gpurun_out/profiles_r06/r06_pipeline_f16x3_pmc.json $O/pmc_pipe_FETCH_SIZE $O/pmc_pipe_WRITE_SIZE $O/pmc_pipe_SQ_VALU_MFMA_BUSY_CYCLES \
  "gemm_f16x3_M196000_N256_K1024=gemm_x3_kernel<256, 2, false, true, true, false>:784384" \
  "conv_f16x3_M196000_N256_K2304=conv_halo_x3_kernel<2, 320, true, true>:784384" \
  "conv_f16x3_M784000_N128_K1152=conv_halo_x3_kernel<2, 320, true, true>:1568256" \
  "conv_f16x3_M49000_N512_K4608=conv_halo_x3_kernel<2, 320, true, true>:393216" \
  "tail_f16x3_c64_cn64_P3136000=chain_x3_kernel<64, true, false, true, false>:12544000" \
  "tail_f16x3_c64_cn64_ds_P3136000=chain_x3_kernel<64, true, true, true, false>:12544000" \
  "tail_f16x3_c64_cn128_P3136000=chain_x3_kernel<128, true, false, true, false>:12544000" \
  "chainw_f16x3_c256_cn256_P196000=chain_x3w_kernel<256, 256, true, false>:784384" \
  "chainw_f16x3_c128_cn128_P784000=chain_x3w_kernel<128, 128, true, false>:3136000" \
  "chainw_f16x3_c128_cn256_P784000=chain_x3w_kernel<128, 256, true, false>:3136000" \
  "dual_f16x3_M784000_N512_K128_K256=gemm_x3_kernel<128, 2, false, true, false, true>:6272000" \
  "dual_f16x3_M196000_N1024_K256_K512=gemm_x3_kernel<128, 2, false, true, false, true>:3137536" \
  "dual_f16x3_M49000_N2048_K512_K1024=gemm_x3_kernel<128, 2, false, true, false, true>:1568768" \
  "conv1_pool_reduce_f16x3=conv1_pool_x3_kernel<true>:131072"
SQ_BENCH_KERNELS=gpurun_out/profiles_r06/r06_pipeline_f16x3_bench_kernels.json python bench.py --no-secondary --no-cpu-baseline --no-accuracy > gpurun_out/profiles_r06/r06_pipeline_f16x3_bench_line.json 2>/dev/null
SQ_BENCH_KERNELS=gpurun_out/profiles_r06/r06_vis_train_bf16_bench_kernels.json python bench.py --workload vis_train --measure-traffic --no-secondary --no-cpu-baseline > gpurun_out/profiles_r06/r06_vis_train_bf16_bench_line.json 2>/dev/null
SQ_BENCH_KERNELS=gpurun_out/profiles_r06/r06_spatial_bf16_bench_kernels.json python bench.py --workload spatial --measure-traffic --no-secondary --no-cpu-baseline > gpurun_out/profiles_r06/r06_spatial_bf16_bench_line.json 2>/dev/null
SQ_BENCH_KERNELS=gpurun_out/profiles_r06/r06_pipeline_uni_bf16_bench_kernels.json python bench.py --workload pipeline --embedder uni --slides 2 --measure-traffic --no-secondary --no-cpu-baseline > gpurun_out/profiles_r06/r06_pipeline_uni_bf16_bench_line.json 2>/dev/null
python bench.py > gpurun_out/profiles_r06/r06_bench_final.json 2> gpurun_out/profiles_r06/r06_bench_final.err
ls -la gpurun_out/profiles_r06
find $O -name "*.csv" -size +5M -delete
