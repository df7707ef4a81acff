cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/names
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/names -- python $R/tools/blaslt_names.py > $R/gpurun_out/names.log 2>&1
cd $R
f=$(ls gpurun_out/names/*/*kernel_stats.csv | head -1)
cut -d, -f1-4 $f | head -20
