#!/bin/bash
# Runs on the GPU box (gpurun): everything profiles/r02_* is derived from, into gpurun_out/r02/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02
rm -rf $O; mkdir -p $O
# 1. the bench line (default flags) and the driver's flags
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_5.json 2> $O/bench_20_5.err
# 2. kernel trace of the bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --no-cpu-baseline --no-pmc > $O/kt.log 2>&1
find $O -name "*kernel_trace.csv" -delete
# 3. hardware counters per kernel (separate passes, counters only)
TPOSE_PMC_GROUPS=0,1,2,3,4,5,7 python tools/pmc_kernels.py $O/pmc.json > /dev/null 2> $O/pmc.err
# 4. in-kernel timelines (debug flavour of the library)
python tools/kernel_timeline.py > $O/timeline.json 2> $O/timeline.err
# 5. other configurations
for cfg in "2048 2048 3000 1" "4096 4096 12000 1" "4096 4096 12000 0" "674 449 150 0" "2048 2048 2 0"; do
  python tools/time_acc.py $cfg >> $O/configs.jsonl 2>> $O/configs.err
done
python tools/time_coarse.py > $O/coarse.jsonl 2>&1
python tools/time_configs.py > $O/time_configs.jsonl 2>&1
python tools/run_config2.py 600 > $O/config2.txt 2>&1
python tools/run_config3.py > $O/config3.txt 2>&1
python tools/run_batch.py --pairs 8 --iters 512 > $O/config4.json 2> $O/config4.err
python tools/e3_band_split.py > $O/e3.json 2> $O/e3.err
ls -la $O
