# rocprofv3 evidence for round 2 (run on the GPU box through gpurun; outputs under gpurun_out/prof_r02/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r02; mkdir -p $O
PIPE="python $R/bench.py --no-secondary --no-cpu-baseline --steps 3 --warmup 1"
TRAIN="python $R/bench.py --workload vis_train --no-secondary --no-cpu-baseline --steps 30 --warmup 3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe -- $PIPE > $O/pipe.log 2>&1
SQ_RESNET_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe_serial -- $PIPE --no-stream > $O/pipe_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- $TRAIN > $O/train.log 2>&1
SQ_BWD_ONE_STREAM=1 SQ_FWD_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_serial -- $TRAIN > $O/train_serial.log 2>&1
PIPE1="python $R/bench.py --no-secondary --no-cpu-baseline --slides 2 --steps 1 --warmup 1"
TRAIN1="python $R/bench.py --workload vis_train --no-secondary --no-cpu-baseline --steps 3 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_pipe_$n -- $PIPE1 > $O/pmc_pipe_$n.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_train_$n -- $TRAIN1 > $O/pmc_train_$n.log 2>&1
done
cd $R
for w in pipe pipe_serial train train_serial; do f=$(ls $O/$w/*/*kernel_stats.csv 2>/dev/null | head -1); echo "$w: $f"; head -4 $f | cut -c1-150; done
# keep only the small summaries (the traces are large)
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
