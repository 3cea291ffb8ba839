mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_persist_sizes.py -m gpu -x -q -k "metric_size or config1") > gpurun_out/r4_t13.log 2>&1; tail -3 gpurun_out/r4_t13.log
for i in 1 2; do
for e in product head; do
unset TPOSE_HIP_LIB
if [ $e = head ]; then export TPOSE_HIP_LIB=$PWD/tpose_amd/variants/libtpose_hip_head.so; fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-cold --no-pmc --no-cpu-baseline --no-extra > gpurun_out/r4_b20.json 2>gpurun_out/r4_b20.err; python -c "
import json; d=json.load(open('gpurun_out/r4_b20.json')); print('$e bench20 ms_per_step', d['ms_per_step'], 'device', d['ms_per_step_device'], 'kern/iter', d['roofline']['us_per_grad_iter'], d['timing'])"; done; done
unset TPOSE_HIP_LIB
python tools/time_variants.py product head
python tools/launch_profile.py | head -8
