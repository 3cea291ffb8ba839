"""Host-timed grad-iter of library variants (tools/build_variants.py) at the metric size: python tools/time_variants.py name ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import sys, time
sys.path.insert(0, %r)
from tpose_amd import capi, synth
img, pts, tris, he, ratio = synth.workload(2048, 2048, 3000, contrast=0.1)
ctx = capi.Context(0, 2048, 2048)
ctx.set_image(capi.IMAGE_A, img); ctx.upload(pts, tris, None)
p = capi.default_params(0); ctx.prepare(p); ctx.iterate(p, 64); ctx.synchronize()
out = []
for steps in (256, 2048, 2048, 2048):
    t0 = time.perf_counter(); ctx.iterate(p, steps); ctx.synchronize()
    out.append("%%d: %%.2f" %% (steps, (time.perf_counter() - t0) / steps * 1e6))
print(" | ".join(out), "| replans", ctx.info(capi.INFO_REPLANS), "given up", ctx.info(capi.INFO_PERSIST_FAILURES), "rows per lane", ctx.info(13))
""" % ROOT
for v in sys.argv[1:]:
    env = dict(os.environ)
    if v != "product":
        env["TPOSE_HIP_LIB"] = os.path.join(ROOT, "tpose_amd", "variants", "libtpose_hip_%s.so" % v)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(v, "us/grad-iter after 64 |", r.stdout.strip() or r.stderr[-800:], flush=True)
