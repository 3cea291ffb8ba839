#!/usr/bin/env python
"""Host-timed grad-iter of library variants at BASELINE config 4's element (4096^2 / 12 000 triangles): python tools/time_big.py name ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import sys, time
sys.path.insert(0, %r)
from tpose_amd import capi, synth
img, pts, tris, he, ratio = synth.workload(4096, 4096, 12000, contrast=0.1)
ctx = capi.Context(0, 4096, 4096)
ctx.set_image(capi.IMAGE_A, img); ctx.upload(pts, tris, None)
p = capi.default_params(0); ctx.prepare(p); ctx.iterate(p, 64); ctx.synchronize()
out = []
for steps in (256, 1024, 1024):
    t0 = time.perf_counter(); ctx.iterate(p, steps); ctx.synchronize()
    out.append("%%d: %%.2f" %% (steps, (time.perf_counter() - t0) / steps * 1e6))
print(" | ".join(out), "| patches", ctx.info(capi.INFO_PATCHES), "LDS", ctx.info(capi.INFO_PATCH_LDS), "replans", ctx.info(capi.INFO_REPLANS))
""" % ROOT
for v in sys.argv[1:]:
    env = dict(os.environ)
    if v != "product":
        env["TPOSE_HIP_LIB"] = os.path.join(ROOT, "tpose_amd", "variants", "libtpose_hip_%s.so" % v)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    print(v, "4096^2/12000 us/grad-iter after 64 |", r.stdout.strip() or r.stderr[-800:], flush=True)
