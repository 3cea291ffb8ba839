mkdir -p gpurun_out
for k in concentrated nan inf; do timeout 120 python tools/hostile_repro.py $k 6 1 > gpurun_out/host_$k.log 2>&1; echo "rc $?" >> gpurun_out/host_$k.log; tail -4 gpurun_out/host_$k.log; done
(timeout 900 python -m pytest tests/test_persist_sizes.py tests/test_bands.py tests/test_hip_parity.py -m gpu -x -q -k "not hostile_vertex_sets_on") > gpurun_out/r4_t2.log 2>&1; tail -3 gpurun_out/r4_t2.log
python tools/time_variants.py base product base product > gpurun_out/r4_tv2.txt 2>&1; cat gpurun_out/r4_tv2.txt
timeout 200 python tools/wave_timeline.py > gpurun_out/r4_wave_c1.json 2> gpurun_out/r4_wave_c1.err; tail -60 gpurun_out/r4_wave_c1.json
timeout 200 python tools/persist_timeline.py > gpurun_out/r4_tl_c1.json 2>/dev/null; head -50 gpurun_out/r4_tl_c1.json
