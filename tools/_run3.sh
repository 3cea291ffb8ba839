mkdir -p gpurun_out
python tools/check_variants.py product > gpurun_out/r4_cv5.txt 2>&1; cat gpurun_out/r4_cv5.txt
for rep in 1 2 3; do timeout 120 python tools/hostile_repro.py concentrated 6 1 --check > gpurun_out/hc_p$rep.log 2>&1; echo "rc $?"; grep -v "^  File" gpurun_out/hc_p$rep.log | grep -i "fault\|equal\|persist" | tail -3; done
python tools/time_variants.py product base product > gpurun_out/r4_tv5.txt 2>&1; cat gpurun_out/r4_tv5.txt
timeout 200 python tools/wave_timeline.py > gpurun_out/r4_wave_c1.json 2> gpurun_out/r4_wave_c1.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_wave_c1.json"))
print("period", d["grad-iter period"])
for k,v in d["intervals"].items(): print("%-28s"%k, v)
for k,v in d["arrival after the workgroup's first wave passed the top"].items(): print("arr %-24s"%k, v)
PY
