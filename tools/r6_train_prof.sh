# rocprofv3 kernel stats of the ViS training step (config 2), serial (helper streams off) and as run; summaries -> gpurun_out/profiles_r06/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06; mkdir -p $O $R/gpurun_out/profiles_r06
TRAIN="python $R/bench.py --workload vis_train --no-secondary --no-cpu-baseline --steps 30 --warmup 3"
rm -rf $O/train $O/train_serial
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- $TRAIN > $O/train.log 2>&1
SQ_BWD_ONE_STREAM=1 SQ_FWD_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_serial -- $TRAIN > $O/train_serial.log 2>&1
cd $R
for w in train train_serial; do
  f=$(ls $O/$w/*/*kernel_stats.csv 2>/dev/null | head -1)
  case $w in train) t=r06_vis_train_bf16_kernel_stats.csv;; train_serial) t=r06_vis_train_bf16_kernel_stats_serial.csv;; esac
  [ -n "$f" ] && cp $f gpurun_out/profiles_r06/$t
done
tail -1 $O/train.log; tail -1 $O/train_serial.log
find $O -name "*.csv" -size +5M -delete
