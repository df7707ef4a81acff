# rocprofv3 evidence for round 4 (run on the GPU box through gpurun; outputs under gpurun_out/prof_r04/, summaries copied to profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r04; rm -rf $O; mkdir -p $O
PIPE="python $R/bench.py --no-secondary --no-cpu-baseline --no-accuracy --steps 3 --warmup 1"
TRAIN="python $R/bench.py --workload vis_train --no-secondary --no-cpu-baseline --steps 30 --warmup 3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe -- $PIPE > $O/pipe.log 2>&1
SQ_RESNET_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe_serial -- $PIPE --no-stream > $O/pipe_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- $TRAIN > $O/train.log 2>&1
SQ_BWD_ONE_STREAM=1 SQ_FWD_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_serial -- $TRAIN > $O/train_serial.log 2>&1
PIPE1="python $R/bench.py --no-secondary --no-cpu-baseline --no-accuracy --slides 2 --steps 1 --warmup 1"
TRAIN1="python $R/bench.py --workload vis_train --no-secondary --no-cpu-baseline --steps 3 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_pipe_$n -- $PIPE1 > $O/pmc_pipe_$n.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_train_$n -- $TRAIN1 > $O/pmc_train_$n.log 2>&1
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_pipe_SQ -- $PIPE1 > $O/pmc_pipe_SQ.log 2>&1
cd $R
mkdir -p gpurun_out/profiles_r04
for w in pipe pipe_serial train train_serial; do
  f=$(ls $O/$w/*/*kernel_stats.csv 2>/dev/null | head -1)
  case $w in pipe) t=r04_pipeline_f16x3_kernel_stats.csv;; pipe_serial) t=r04_pipeline_f16x3_kernel_stats_serial.csv;; train) t=r04_vis_train_bf16_kernel_stats.csv;; train_serial) t=r04_vis_train_bf16_kernel_stats_serial.csv;; esac
  [ -n "$f" ] && cp $f gpurun_out/profiles_r04/$t
done
python tools/pmc_summary.py r04_pipeline_f16x3 gpurun_out/profiles_r04/r04_pipeline_f16x3_pmc.json $O/pmc_pipe_FETCH_SIZE $O/pmc_pipe_WRITE_SIZE $O/pmc_pipe_SQ_VALU_MFMA_BUSY_CYCLES \
  "gemm_f16x3_M196000_N1024_K256=gemm_x3_kernel<128, 2, false, true, false, false>:3137536" \
  "gemm_f16x3_M784000_N512_K128=gemm_x3_kernel<128, 2, false, true, false, false>:6272000" \
  "gemm_f16x3_M196000_N256_K1024=gemm_x3_kernel<256, 2, false, true, true, false>:784384" \
  "conv_f16x3_M196000_N256_K2304=conv_halo_x3_kernel<2, 320, true, true>:784384" \
  "conv_f16x3_M784000_N128_K1152=conv_halo_x3_kernel<2, 320, true, true>:1568256" \
  "conv_f16x3_M49000_N512_K4608=conv_halo_x3_kernel<2, 320, true, true>:393216" \
  "tail_f16x3_c64_cn64_P3136000=chain_x3_kernel<64, true, false, true>:12544000" \
  "tail_f16x3_c64_cn64_ds_P3136000=chain_x3_kernel<64, true, true, true>:12544000" \
  "tail_f16x3_c64_cn128_P3136000=chain_x3_kernel<128, true, false, true>:12544000" \
  "dual_f16x3_M784000_N512_K128_K256=gemm_x3_kernel<128, 2, false, true, false, true>:6272000" \
  "dual_f16x3_M196000_N1024_K256_K512=gemm_x3_kernel<128, 2, false, true, false, true>:3137536" \
  "conv1_pool_f16x3=conv1_pool_x3_kernel<true>:131072"
python tools/pmc_summary.py r04_vis_train_bf16 gpurun_out/profiles_r04/r04_vis_train_bf16_pmc.json $O/pmc_train_FETCH_SIZE $O/pmc_train_WRITE_SIZE $O/pmc_train_SQ_VALU_MFMA_BUSY_CYCLES \
  "gemm_bf16_M6400_N1024_K1024_b1=gemm_nt_kernel<unsigned short, 2, 2, false:102400" \
  "gemmtn_bf16_M1024_N1024_K6400_b4=gemm_tn_ring_kernel:131072"
python - <<'PY'
import csv, glob, collections, re, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/prof_r04/pmc_pipe_SQ/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))
        acc[n + "|grid=" + r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in acc.items():
    if 'x3' not in k and 'chain' not in k: continue
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    wc = c.get('SQ_WAVE_CYCLES', 1.0)
    out[k] = {"dispatches": len(next(iter(cs.values()))), "SQ_WAVE_CYCLES": round(wc),
              **{n.replace('SQ_', '') + "_frac": round(v / wc, 4) for n, v in sorted(c.items()) if n != 'SQ_WAVE_CYCLES'}}
json.dump({"note": "SQ counters as fractions of SQ_WAVE_CYCLES per kernel symbol and grid size (threads); bench.py --slides 2 --steps 1 --warmup 1 (f16x3 pipeline); WAIT_ANY = parked at s_waitcnt / s_barrier, WAIT_INST_ANY = issue stalls, ACTIVE_INST_ANY = issuing", "kernels": out},
          open('gpurun_out/profiles_r04/r04_pipeline_f16x3_sq_counters.json', 'w'), indent=1)
print("sq kernels", len(out))
PY
SQ_BENCH_KERNELS=gpurun_out/profiles_r04/r04_pipeline_f16x3_bench_kernels.json python bench.py --no-secondary --no-cpu-baseline --no-accuracy > gpurun_out/profiles_r04/r04_pipeline_f16x3_bench_line.json 2>/dev/null
python bench.py > gpurun_out/profiles_r04/r04_bench_final.json 2>/dev/null
SQ_BENCH_KERNELS=gpurun_out/profiles_r04/r04_pipeline_uni_bf16_bench_kernels.json python bench.py --workload pipeline --embedder uni --slides 2 --no-secondary --no-cpu-baseline > gpurun_out/profiles_r04/r04_pipeline_uni_bf16_bench_line.json 2>/dev/null
SQ_BENCH_KERNELS=gpurun_out/profiles_r04/r04_spatial_bf16_bench_kernels.json python bench.py --workload spatial --no-secondary --no-cpu-baseline > gpurun_out/profiles_r04/r04_spatial_bf16_bench_line.json 2>/dev/null
cd /tmp; for w in "uni:--workload pipeline --embedder uni --slides 1 --steps 2 --warmup 1" "spatial:--workload spatial --steps 1 --warmup 1"; do
  t=${w%%:*}; a=${w#*:}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$t -- python $R/bench.py --no-secondary --no-cpu-baseline $a > $O/$t.log 2>&1
  f=$(ls $O/$t/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/profiles_r04/r04_${t}_bf16_kernel_stats.csv
done
cd $R
bash tools/r4_tnpmc.sh > /dev/null 2>&1; cp gpurun_out/r4_tn_sq_counters.txt gpurun_out/profiles_r04/r04_tn_sq_counters.txt
python tools/tn_probe.py group 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_r04/r04_tn_probe_ring.txt
python tools/nt_small.py 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_r04/r04_nt_small_cold_operands.txt
SQ_BENCH_KERNELS=gpurun_out/profiles_r04/r04_vis_train_bf16_bench_kernels.json python bench.py --workload vis_train --no-secondary --no-cpu-baseline > gpurun_out/profiles_r04/r04_vis_train_bf16_bench_line.json 2>/dev/null
ls -la gpurun_out/profiles_r04
find $O -name "*.csv" -size +5M -delete
du -sh $O
