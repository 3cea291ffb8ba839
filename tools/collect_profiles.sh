#!/bin/bash
# Runs on the GPU box (gpurun): everything profiles/ is derived from, into gpurun_out/final/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --no-cpu-baseline --no-pmc > $O/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-pmc > $O/pmc_$c.log 2>&1
done
python tools/time_configs.py > $O/configs.jsonl 2> $O/configs.err
python tools/acc_timeline.py > $O/acc_timeline.json 2> $O/acc_timeline.err
python tools/time_coarse.py > $O/coarse.jsonl 2>&1
python tools/probe_launch.py > $O/launch_probe.txt 2>&1
python tools/run_config2.py 600 > $O/config2.txt 2>&1
python tools/run_config3.py > $O/config3.txt 2>&1
python tools/run_batch.py --pairs 8 --iters 512 > $O/config4.json 2> $O/config4.err
# keep the merged directory small: only the per-kernel summaries of the traces
find $O -name "*kernel_trace.csv" -size +30M -delete
ls -la $O $O/kt/* | head -40
