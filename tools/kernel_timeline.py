#!/usr/bin/env python
"""In-kernel timelines of the two kernels of a fused grad-iter at the headline workload.  Uses the DEBUG flavour of the
library (built here with -DTPOSE_DEBUG into tpose_amd/variants/; the product library has no such hooks): thread 0 of
every workgroup stamps the 100 MHz wall clock at its phase boundaries.  Needs an MI355X.  Prints one JSON object:
per kernel, when workgroups start / reach each stamp / end, in microseconds after the kernel's first stamp."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_path = os.environ.get("TPOSE_TIMELINE_LIB") or os.path.join(ROOT, "tpose_amd", "variants", "libtpose_hip_debug.so")
os.environ["TPOSE_HIP_LIB"] = lib_path  # before tpose_amd.capi is imported
from tpose_amd import build as tb  # noqa: E402

if not os.path.exists(lib_path) or "--rebuild" in sys.argv:
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    tb.build(force=True, extra=["-DTPOSE_DEBUG"], out=lib_path)
import numpy as np  # noqa: E402

from tpose_amd import capi, synth  # noqa: E402

W = H = 2048
NT = 3000
img, pts, tris, he, ratio = synth.workload(W, H, NT)
ctx = capi.Context(0, W, H)
ctx.set_image(capi.IMAGE_A, img)
ctx.upload(pts, tris, None)
p = capi.default_params(capi.TRIANGULATE)
ctx.iterate(p, 8)
ctx.synchronize()
lib = ctx.lib
lib.tp_debug_dump.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
NB = 4096
names = {0: ("k_lines", ["start", "lines set up", "rows walked", "end"]),
         2: ("k_update", ["start", "line sums", "variants", "end"])}
runs = []
for rep in range(24):
    ctx.iterate(p, 1)
    ctx.synchronize()
    buf = np.zeros(3 * NB * 8, np.uint64)
    assert lib.tp_debug_dump(ctx.h, buf.ctypes.data, buf.size) == 0
    runs.append(buf.reshape(3, NB, 8).astype(np.int64))
st = np.stack(runs)  # [launch, region, block, stamp]
out = {"units": "us after the first workgroup's first stamp of the same launch; percentiles over 24 launches x workgroups"}
for reg, (name, labels) in names.items():
    a = st[:, reg]
    used = a[:, :, 0] > 0
    t0 = np.where(used, a[:, :, 0], np.iinfo(np.int64).max).min(axis=1)[:, None, None]
    rel = (a - t0) / 100.0
    d = {}
    for k, lab in enumerate(labels):
        v = rel[:, :, k][used & (a[:, :, k] > 0)]
        if v.size:
            d[lab] = {str(q): round(float(np.percentile(v, q)), 2) for q in (1, 50, 90, 100)}
    d["workgroups stamped"] = int(used[0].sum())
    out[name] = d
print(json.dumps(out, indent=1))
