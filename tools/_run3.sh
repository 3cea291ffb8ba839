mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -x -q -k "until or harness or warp or config") > gpurun_out/r4_t15.log 2>&1; tail -3 gpurun_out/r4_t15.log
timeout 600 python bench.py --no-cold --no-pmc --no-cpu-baseline > gpurun_out/r4_bdef.json 2>gpurun_out/r4_bdef.err; python -c "
import json; d=json.load(open('gpurun_out/r4_bdef.json')); print('bench ms_per_step', d['ms_per_step'], 'full', d.get('ms_per_step_full_contrast'), 'all13', d.get('ms_per_step_all_13_variants'))"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cold --no-pmc --no-cpu-baseline > gpurun_out/r4_b20.json 2>gpurun_out/r4_b20.err; python -c "
import json; d=json.load(open('gpurun_out/r4_b20.json')); print('bench20 ms_per_step', d['ms_per_step'], 'full', d.get('ms_per_step_full_contrast'), 'all13', d.get('ms_per_step_all_13_variants'))"
python tools/run_config3.py 2>&1 | tail -5
