mkdir -p gpurun_out
python tools/check_variants.py product > gpurun_out/r4_cv10.txt 2>&1; cat gpurun_out/r4_cv10.txt
python tools/time_variants.py product base product > gpurun_out/r4_tv10.txt 2>&1; cat gpurun_out/r4_tv10.txt
(timeout 900 python -m pytest tests/test_bands.py tests/test_persist_sizes.py tests/test_hip_parity.py -m gpu -x -q) > gpurun_out/r4_t6.log 2>&1; tail -2 gpurun_out/r4_t6.log
