mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -x -q -k "evaluate or harness or config") > gpurun_out/r4_t16.log 2>&1; tail -5 gpurun_out/r4_t16.log
python tools/run_config2.py 600 2>&1 | tail -6
python tools/run_config2.py 600 2>&1 | tail -6
