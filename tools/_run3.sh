python tools/run_config2.py 600 2>&1 | tail -5
python tools/run_config2.py 600 2>&1 | tail -5
(timeout 600 python -m pytest tests -m gpu -x -q -k "config2 or harness or evaluate") 2>&1 | tail -3
