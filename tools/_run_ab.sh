export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r2p; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/ -q -m gpu 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
python tools/time_acc.py > $O/time.json 2>$O/time.err; cat $O/time.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --no-cpu-baseline --no-pmc --steps 512 --warmup 64 > $O/kt.log 2>&1
find $O -name "*kernel_trace.csv" -delete
head -5 $O/kt/kt_kernel_stats.csv | cut -c1-150
