export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r2q; mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_5.json 2> $O/bench_20_5.err; python -c "
import json; d=json.load(open('$O/bench_20_5.json')); print('20/5:', d['value'], d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['kernel_us_hip_events'], d['roofline']['frac'], d['roofline']['traffic'])"
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default:', d['value'], d['ms_per_step'], d['timing'], d['roofline']['kernel_us'], d['roofline']['kernel_us_hip_events'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['kernel_trace_us'], d['cpu_baseline'])"
tail -3 $O/bench_default.err
