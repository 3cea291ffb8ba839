mkdir -p gpurun_out
python tools/time_big.py product ub4 > gpurun_out/r4_big2.txt 2>&1; cat gpurun_out/r4_big2.txt
python tools/time_variants.py product ub4 > gpurun_out/r4_tv13.txt 2>&1; cat gpurun_out/r4_tv13.txt
(timeout 600 python -m pytest tests/test_persist_sizes.py -m gpu -x -q -k "batch_size or config1") > gpurun_out/r4_t7.log 2>&1; tail -2 gpurun_out/r4_t7.log
