mkdir -p gpurun_out
for i in 1 2; do
python tools/run_batch.py --pairs 8 --iters 512 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('product', d['triangles_iters_per_s'], d['seconds_iterating_slowest_rank'], [round(p['seconds'],4) for p in d['pairs']], [p['launches_given_up'] for p in d['pairs']])"
done
(timeout 900 python -m pytest tests/test_bands.py tests/test_stress.py tests/test_harness.py -m gpu -x -q) > gpurun_out/r4_t14.log 2>&1; tail -3 gpurun_out/r4_t14.log
python tools/thread_stress.py 2>&1 | tail -3
