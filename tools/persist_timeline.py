#!/usr/bin/env python
"""In-kernel timeline of the persistent grad-iter kernel at the headline workload (debug flavour of the library, built
here with -DTPOSE_DEBUG into tpose_amd/variants/): thread 0 of every workgroup stamps the 100 MHz wall clock at the
phase boundaries of the first 64 grad-iters of a launch.  Needs an MI355X.  Prints one JSON object: per phase, how long
workgroups spend in it (microseconds; percentiles over workgroups x grad-iters 8..63), and the grad-iter period."""
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

args = [a for a in sys.argv[1:] if not a.startswith("--")]
W = H = int(args[0]) if args else 2048
NT = int(args[1]) if len(args) > 1 else 3000
contrast = float(os.environ.get("TPOSE_CONTRAST", "0.1"))
img, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=contrast)
ctx = capi.Context(0, W, H)
ctx.set_image(capi.IMAGE_A, img)
ctx.upload(pts, tris, None)
p = capi.default_params(capi.TRIANGULATE)
# TPOSE_TIMELINE_AFTER=n: n grad-iters first (their launches stamp too; the last launch's stamps are the ones read)
after = int(os.environ.get("TPOSE_TIMELINE_AFTER", "0"))
first = after + int(os.environ.get("TPOSE_DBG_FIRST", "0"))   # (TPOSE_DBG_FIRST: first stamped grad-iter of a launch, read by the library)
if after:
    ctx.iterate(p, after)
ctx.iterate(p, int(os.environ.get("TPOSE_DBG_FIRST", "0")) + 130)
ctx.synchronize()
lib = ctx.lib
lib.tp_debug_dump_persist.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
parts = ctx.info(capi.INFO_PATCHES)
IT = 64
buf = np.zeros(512 * IT * 16, np.uint64)
assert lib.tp_debug_dump_persist(ctx.h, buf.ctypes.data, buf.size) == 0
st = buf.reshape(512, IT, 16)[:parts].astype(np.int64)  # [workgroup, grad-iter, stamp]
labels = ["P0 positions in (wait)", "P1 set-up + snap", "P3 walk", "P6 corners", "P7 step + post"]
out = {"workload": "%dx%d / %d triangles, contrast %g, grad-iters %d.." % (W, H, tris.shape[0], contrast, first + 8), "patches": parts,
       "units": "us; percentiles over workgroups x grad-iters 8..63 of one launch"}
sel = st[:, 8:IT]
for k, lab in enumerate(labels):
    d = (sel[:, :, k + 1] - sel[:, :, k]) / 100.0
    out[lab] = {str(q): round(float(np.percentile(d, q)), 2) for q in (1, 50, 90, 99, 100)}
# inside P3, thread 0: columns scanned and stale records requested for both lane-items; records summed and folded into LDS
sub = [("P3a scan + fetch", 2, 8), ("P3b sums + atomics", 8, 9), ("P3e barrier", 9, 3)]
for lab, a, b in sub:
    d = (sel[:, :, b] - sel[:, :, a]) / 100.0
    ok = (sel[:, :, a] > 0) & (sel[:, :, b] > 0)
    if ok.any():
        out[lab] = {str(q): round(float(np.percentile(d[ok], q)), 2) for q in (1, 50, 90, 100)}
# the first grad-iters of the launch (prologue behind them: tables into LDS, the first cut of the lines, every record fetched)
for it0 in (0, 1, 2):
    out["grad-iter %d of the launch, median per phase" % it0] = {lab.split()[0]: round(float(np.median((st[:, it0, k + 1] - st[:, it0, k]) / 100.0)), 2)
                                                                  for k, lab in enumerate(labels)}
out["first stamp to start of grad-iter 8, median"] = round(float(np.median((st[:, 8, 0] - st[:, 0, 0]) / 100.0)), 2)
period = (st[:, IT - 1, 0] - st[:, 8, 0]) / 100.0 / (IT - 1 - 8)
out["grad-iter period"] = {"mean": round(float(period.mean()), 3), "min": round(float(period.min()), 3), "max": round(float(period.max()), 3)}
start = (st[:, 0, 0] - st[:, 0, 0].min()) / 100.0
out["workgroup start spread"] = round(float(start.max()), 2)
print(json.dumps(out, indent=1))
