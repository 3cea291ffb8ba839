# timing variants of the persistent kernel (tools/build_variants.py) through tools/persist_timeline.py; one line each
mkdir -p gpurun_out
for v in "$@"; do
  TPOSE_TIMELINE_LIB=$PWD/tpose_amd/variants/libtpose_hip_$v.so timeout 120 python tools/persist_timeline.py > gpurun_out/tl_$v.json 2>gpurun_out/tl_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/tl_$v.json"))
    print("$v", "period", d["grad-iter period"]["mean"], "| " + " | ".join("%s %s/%s/%s" % (k.split()[0], v["1"], v["50"], v["100"]) for k,v in d.items() if k.startswith("P") and isinstance(v, dict)))
except Exception as e:
    print("$v", "FAILED", e); print(open("gpurun_out/tl_$v.err").read()[-2000:])
PY
done
