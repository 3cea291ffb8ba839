#!/usr/bin/env python
"""Per-WAVE in-kernel timeline of the persistent grad-iter kernel (debug flavour of the library, -DTPOSE_DEBUG): the first lane
of every wave stamps the 100 MHz wall clock where it arrives at / leaves every workgroup barrier of a grad-iter.  Shows which
waves are the long pole of each phase and how long the others wait for them.  Needs an MI355X.  Prints one JSON object:
per stamp interval and wave, the median over workgroups x grad-iters 8..31 (microseconds).
  python tools/wave_timeline.py [W NT]         (TPOSE_TIMELINE_LIB picks a variant library)"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_path = os.environ.get("TPOSE_TIMELINE_LIB") or os.path.join(ROOT, "tpose_amd", "variants", "libtpose_hip_debugwaves.so")
os.environ["TPOSE_HIP_LIB"] = lib_path
from tpose_amd import build as tb  # noqa: E402

if not os.path.exists(lib_path) or "--rebuild" in sys.argv:
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    tb.build(force=True, extra=["-DTPOSE_DEBUG", "-DPK_DBG_WAVES"], out=lib_path)
import numpy as np  # noqa: E402

from tpose_amd import capi, synth  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
W = H = int(args[0]) if args else 2048
NT = int(args[1]) if len(args) > 1 else 3000
WAVES = int(os.environ.get("TPOSE_WAVES", "8"))
from tpose_amd import photos  # noqa: E402
img, pts, tris, he, ratio, raster_label = photos.raster_from_env(W, H, NT)
ctx = capi.Context(0, W, H)
ctx.set_image(capi.IMAGE_A, img)
ctx.upload(pts, tris, None)
p = capi.default_params(capi.TRIANGULATE)
ctx.iterate(p, 256)
ctx.iterate(p, 130)
ctx.synchronize()
lib = ctx.lib
lib.tp_debug_dump_persist.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
parts = ctx.info(capi.INFO_PATCHES)
IT, WIT = 64, 32
base = 512 * IT * 16
buf = np.zeros(base + 512 * WIT * 16 * 16, np.uint64)
assert lib.tp_debug_dump_persist(ctx.h, buf.ctypes.data, buf.size) == 0
st = buf[base: base + parts * WIT * WAVES * 16].reshape(parts, WIT, WAVES, 16).astype(np.int64)
names = os.environ.get("TPOSE_STAMPS", "top,P0 polled,P0 barrier,P1 done,P1 barrier,P3 pass,P3 folded,P3 barrier,P6 done,P6 barrier,P7 end").split(",")
out = {"workload": "%dx%d / %d triangles, %s" % (W, H, tris.shape[0], raster_label), "patches": parts, "waves": WAVES,
       "units": "us, median over workgroups x grad-iters 8..31; per wave", "intervals": {}}
sel = st[:, 8:WIT]
for k in range(len(names) - 1):
    a, b = sel[:, :, :, k], sel[:, :, :, k + 1]
    ok = (a > 0) & (b > 0)
    row = []
    for w in range(WAVES):
        d = (b[:, :, w] - a[:, :, w])[ok[:, :, w]] / 100.0
        row.append(round(float(np.median(d)), 2) if d.size else None)
    out["intervals"]["%s -> %s" % (names[k], names[k + 1])] = row
# when, after the workgroup's first wave passed `top`, each wave arrives at each stamp (median)
t0 = sel[:, :, :, 0].min(axis=2)[:, :, None]
arr = {}
for k in range(len(names)):
    row = []
    for w in range(WAVES):
        v = sel[:, :, w, k]
        d = ((v - t0[:, :, 0])[v > 0]) / 100.0
        row.append(round(float(np.median(d)), 2) if d.size else None)
    arr[names[k]] = row
out["arrival after the workgroup's first wave passed the top"] = arr
period = (st[:, WIT - 1, 0, 0] - st[:, 8, 0, 0]) / 100.0 / (WIT - 1 - 8)
out["grad-iter period"] = round(float(period.mean()), 3)
print(json.dumps(out, indent=1))
