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
from tpose_amd import photos  # noqa: E402
img, pts, tris, he, ratio, raster_label = photos.raster_from_env(W, H, NT)
ctx = capi.Context(0, W, H)
ctx.set_image(capi.IMAGE_A, img)
ctx.upload(pts, tris, None)
p = capi.default_params(capi.TRIANGULATE)
# TPOSE_TIMELINE_AFTER=n: n grad-iters first (their launches stamp too; the last launch's stamps are the ones read)
after = int(os.environ.get("TPOSE_TIMELINE_AFTER", "0"))
first = after + int(os.environ.get("TPOSE_DBG_FIRST", "0"))   # (TPOSE_DBG_FIRST: first stamped grad-iter of a launch, read by the library)
if os.environ.get("TPOSE_TIMELINE_PREPARE"):   # (the bench's shape: tp_prepare first -- the plan with the probe's speeds)
    ctx.prepare(p)
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
if os.environ.get("TPOSE_TIMELINE_DUMP"):   # the stamps themselves, [workgroup][grad-iter][16] (100 MHz ticks), for tools that want them per patch
    np.save(os.environ["TPOSE_TIMELINE_DUMP"], st)
labels = ["P0 positions in (wait)", "P1 set-up + snap", "P3 walk", "P6 corners", "P7 step + post"]
out = {"workload": "%dx%d / %d triangles, %s, grad-iters %d.." % (W, H, tris.shape[0], raster_label, first + 8), "patches": parts,
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
# who sets the pace: per workgroup, the median over grad-iters of its own chain (P1 + P3 + P6: everything but the wait for positions)
chain = (sel[:, :, 1:2] * 0 + (sel[:, :, 5] - sel[:, :, 1])[:, :, None])[:, :, 0] / 100.0   # stamp 1 (positions in) -> stamp 5 (end of the grad-iter)
cm = np.median(chain, axis=1)
p3 = np.median((sel[:, :, 3] - sel[:, :, 2]) / 100.0, axis=1)
order = np.argsort(-cm)
out["chain P1..P7 per workgroup, median over grad-iters"] = {str(q): round(float(np.percentile(cm, q)), 2) for q in (0, 10, 50, 90, 100)}
out["P3 per workgroup, median over grad-iters"] = {str(q): round(float(np.percentile(p3, q)), 2) for q in (0, 10, 50, 90, 100)}
p3a = np.median((sel[:, :, 8] - sel[:, :, 2]) / 100.0, axis=1); p3b = np.median((sel[:, :, 9] - sel[:, :, 8]) / 100.0, axis=1); p3e = np.median((sel[:, :, 3] - sel[:, :, 9]) / 100.0, axis=1)
p1 = np.median((sel[:, :, 2] - sel[:, :, 1]) / 100.0, axis=1); p6 = np.median((sel[:, :, 5] - sel[:, :, 3]) / 100.0, axis=1)
out["slowest workgroups (block: chain, P1, P3 = scan + sums + barrier, P6)"] = {str(int(b)): [round(float(v[b]), 2) for v in (cm, p1, p3, p3a, p3b, p3e, p6)] for b in order[:12]}
out["fastest workgroups (block: chain, P1, P3 = scan + sums + barrier, P6)"] = {str(int(b)): [round(float(v[b]), 2) for v in (cm, p1, p3, p3a, p3b, p3e, p6)] for b in order[-6:]}
try:
    pp = np.zeros((512, 8), np.int32)
    # (what the plan says about the slowest: own vertices, slots, edges, lines, lane-items, corners, rows per lane -- CPU replay of the planner)
    import subprocess
    so = os.path.join(ROOT, "tests", "_build", "libtp_emul_persist_tl.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-o", so, os.path.join(ROOT, "tests", "emul", "emul_persist.cpp")])
    emp = ctypes.CDLL(so)
    dp_px = 0.05 / (1 + 4 * tris.shape[0] / 3000.0) * H / 2
    n = emp.emul_plan_patches(pts.ctypes.data_as(ctypes.c_void_p), pts.shape[0], tris.ctypes.data_as(ctypes.c_void_p), tris.shape[0], W, H, ctypes.c_float(ratio), ctypes.c_float(dp_px),
                              parts, 160 * 1024 - 512, pp.ctypes.data_as(ctypes.c_void_p), 512)
    def patch_of_block(b): return (b & 7) * (parts >> 3) + (b >> 3) if parts % 8 == 0 else b
    out["plan of the slowest (block: own vertices, slots, edges, lines, lane-items, corners, rows per lane)"] = {str(int(b)): [int(x) for x in pp[patch_of_block(int(b))][:7]] for b in order[:12]}
    out["plan of the fastest"] = {str(int(b)): [int(x) for x in pp[patch_of_block(int(b))][:7]] for b in order[-6:]}
except Exception as e:  # noqa: BLE001
    out["plan of the slowest"] = "unavailable: %s" % e
# P3 of every grad-iter, max over workgroups: spikes are the grad-iters in which lines are cut again
out["P3 max over workgroups by grad-iter"] = [round(float(v), 2) for v in ((sel[:, :, 3] - sel[:, :, 2]) / 100.0).max(axis=0)]
print(json.dumps(out, indent=1))
