python tools/wave_timeline.py --rebuild > gpurun_out/r04/wave_timeline.json 2> gpurun_out/wave.err; tail -3 gpurun_out/wave.err; wc -c gpurun_out/r04/wave_timeline.json
(timeout 600 python -m pytest tests/test_persist_sizes.py -m gpu -x -q -k "take_turns") 2>&1 | tail -3
