cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py 2>&1 | tail -1 > gpurun_out/bench_r01.json; python -c "
import json; d=json.load(open('gpurun_out/bench_r01.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -o r01 -- python bench.py --no-cpu-baseline > gpurun_out/prof_r01.log 2>&1
head -6 gpurun_out/prof_r01/r01_kernel_stats.csv
python -c "
import json; d=json.loads(open('gpurun_out/prof_r01.log').read().strip().splitlines()[-1]); print('under rocprof: kernel_us', d['roofline']['kernel_us'])"
