cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -o r1 -- python bench.py --steps 512 --warmup 64 --no-cpu-baseline > gpurun_out/prof_r1.log 2>&1
f=$(find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -1); cat "$f" | head -5
