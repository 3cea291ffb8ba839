mkdir -p gpurun_out
python tools/check_variants.py product > gpurun_out/r4_cv8.txt 2>&1; cat gpurun_out/r4_cv8.txt
python tools/time_variants.py product base product > gpurun_out/r4_tv8.txt 2>&1; cat gpurun_out/r4_tv8.txt
python bench.py --steps 20 --warmup 5 --no-cold --no-pmc --no-cpu-baseline > gpurun_out/r4_b20.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r4_b20.json')); print('bench20 ms_per_step', d['ms_per_step'], 'device', d['ms_per_step_device'], 'kern/iter', d['roofline']['us_per_grad_iter'], d['timing'])"
(timeout 900 python -m pytest tests/test_persist_sizes.py tests/test_hip_parity.py tests/test_configs.py -m gpu -x -q) > gpurun_out/r4_t4.log 2>&1; tail -2 gpurun_out/r4_t4.log
