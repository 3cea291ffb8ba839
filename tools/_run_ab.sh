export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3c; mkdir -p $O
cd $GRAFT_REPO_ROOT
for cfg in "2048 2048 2 0" ; do
TPOSE_TIME_ACC_TORCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python tools/time_acc.py $cfg > $O/kt.log 2>&1
find $O -name "*kernel_trace.csv" -delete
head -8 $O/kt/kt_kernel_stats.csv | cut -c1-150
done
