mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4_bdef.json 2>gpurun_out/r4_bdef.err; python -c "
import json; d=json.load(open('gpurun_out/r4_bdef.json')); r=d['roofline']; print('bench ms_per_step', d['ms_per_step'], 'frac', r['frac'], r['us_per_grad_iter'], r['traffic'], r['kernel_timing'])"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_b20.json 2>gpurun_out/r4_b20.err; python -c "
import json; d=json.load(open('gpurun_out/r4_b20.json')); r=d['roofline']; print('bench20 ms_per_step', d['ms_per_step'], 'frac', r['frac'], r['us_per_grad_iter'], r['traffic'])"
