mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/r4_gputest5.log 2>&1; tail -3 gpurun_out/r4_gputest5.log
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cold --no-pmc --no-cpu-baseline > gpurun_out/r4_b20.json 2>gpurun_out/r4_b20.err; python -c "
import json; d=json.load(open('gpurun_out/r4_b20.json')); print('bench20 ms_per_step', d['ms_per_step'], 'device', d['ms_per_step_device'], 'kern/iter', d['roofline']['us_per_grad_iter'], d['timing'], d.get('ms_per_step_full_contrast'), d.get('ms_per_step_all_13_variants'))"; done
timeout 300 python tools/call_length.py 2>&1 | head -3
