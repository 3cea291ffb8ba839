#!/usr/bin/env python
"""Bit-exactness of library variants (tools/build_variants.py) at the metric size: 12 + 5 grad-iters of the persistent kernel
against the oracle, one subprocess per variant.  python tools/check_variants.py name ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import sys
sys.path.insert(0, %r)
import numpy as np
from oracle import oracle as O
from tpose_amd import capi, synth
img, pts, tris, he, ratio = synth.workload(2048, 2048, 3000, contrast=0.1)
ctx = capi.Context(0, 2048, 2048)
ctx.set_image(capi.IMAGE_A, img); ctx.upload(pts, tris, None)
p = capi.default_params(0); ctx.prepare(p); ctx.iterate(p, 12); ctx.synchronize()
ref = O.iterate(img, pts, tris, 0, ratio, 0.00005, 12, literal=False)
ok = [np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"]), np.array_equal(ctx.retrieve(capi.BUF_GRADIENT), ref["gr"]),
      np.array_equal(ctx.retrieve(capi.BUF_POINTS).view(np.uint32), ref["points"].view(np.uint32))]
print("tenergy %%s gradient %%s points %%s | persist iters %%d failures %%d" %% (ok[0], ok[1], ok[2], ctx.info(6), ctx.info(9)))
""" % ROOT
for v in sys.argv[1:]:
    env = dict(os.environ)
    if v != "product":
        env["TPOSE_HIP_LIB"] = os.path.join(ROOT, "tpose_amd", "variants", "libtpose_hip_%s.so" % v)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(v, "|", r.stdout.strip() or r.stderr[-800:], flush=True)
