mkdir -p gpurun_out
(time timeout 900 python -m pytest tests/test_stress.py -m gpu -x -q -s) > gpurun_out/r4_stress.log 2>&1; grep -E "stress:|passed|failed|Error|assert" gpurun_out/r4_stress.log | head; tail -3 gpurun_out/r4_stress.log
(timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench20_b.json 2> gpurun_out/r4_bench20_b.err); python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_bench20_b.json"))
for k in ("value","ms_per_step","ms_per_step_device","ms_per_step_full_contrast","ms_per_step_all_13_variants","ms_per_step_two_kernel_path"): print(k, d[k])
r=d["roofline"]; print({k:r[k] for k in ("bound","frac","us_per_grad_iter","traffic","frac_of_measured_traffic")}); print(d["cold_cache"]); print(d["cpu_baseline"]["value"])
PY
