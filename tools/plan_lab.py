#!/usr/bin/env python
"""Offline: what would the patches' loads be under other vertex weights?  Reads a dump of tools/stale_counts.py (TPOSE_COUNT_DUMP: positions, what every
vertex's lines really fetched and walked per grad-iter, its patch), partitions the same mesh with the planner's bisection + refinement (tools/lab/plan_lab.cpp)
under weights a x rows + b x stale (measured) and under the planner's own, and prints per variant the heaviest patch's stale rows and rows against the means
-- with the fit of experiment 28, P3 = 2.5 us + 0.31 ns x stale rows, the walk of the slowest patch.   python tools/plan_lab.py dump.npz [train.npz]
(train.npz: take the per-vertex measurements from another window -- e.g. the probe's -- and judge them on dump.npz's)"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "tests", "_build", "libplan_lab.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tools", "lab", "plan_lab.cpp"), "-o", so])
lab = C.CDLL(so)
d = np.load(sys.argv[1]); tr = np.load(sys.argv[2]) if len(sys.argv) > 2 else d
pts = np.ascontiguousarray(d["pts0"], np.float32); tris = np.ascontiguousarray(d["tris"], np.int32)
NP, NT = pts.shape[0], tris.shape[0]; W, H, ratio = int(d["W"]), int(d["H"]), float(d["ratio"])
vstale, vrows, owner0 = d["vstale"], d["vrows"], d["owner"]
parts = int(owner0.max()) + 1
def partition(wv, passes=12):
    wv = np.ascontiguousarray(wv, np.float64); out = np.zeros(NP, np.int32)
    lab.lab_partition(NP, NT, tris.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p), W, H, C.c_float(ratio), wv.ctypes.data_as(C.c_void_p), parts, passes, out.ctypes.data_as(C.c_void_p))
    return out
def loads(owner):
    ok = owner >= 0
    return np.bincount(owner[ok], vstale[ok], parts), np.bincount(owner[ok], vrows[ok], parts), np.bincount(owner[ok], minlength=parts)
def report(name, owner):
    s, r, n = loads(owner)
    print("%-44s stale rows per patch: mean %6.0f max %6.0f (x%.2f) | rows: mean %6.0f max %6.0f | vertices min %d max %d | walk of the slowest by the fit: %.2f us (mean %.2f)"
          % (name, s.mean(), s.max(), s.max() / s.mean(), r.mean(), r.max(), n.min(), n.max(), 2.5 + 3.1e-4 * s.max(), 2.5 + 3.1e-4 * s.mean()))
print("%s: %d vertices, %d patches; measured per grad-iter: %.0f rows walked, %.0f fetched again (%.0f %%); heaviest vertex: %.0f stale rows = %.2f of a patch's mean"
      % (sys.argv[1], NP, parts, vrows.sum(), vstale.sum(), 100 * vstale.sum() / vrows.sum(), vstale.max(), vstale.max() / (vstale.sum() / parts)))
report("the plan that ran", owner0)
wv0 = np.zeros(NP, np.float64)
lab.lab_vertex_work(NP, NT, tris.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p), H, None, wv0.ctypes.data_as(C.c_void_p))
report("planner, rows only", partition(wv0))
ts, trw = tr["vstale"], tr["vrows"]
for b in (0.0, 1.0, 1.8, 3.0, 6.0, 12.0, 1e6):
    report("weights rows + %g x stale (measured%s)" % (b, ", other window" if tr is not d else ""), partition(trw + b * ts + 40.0))
for b in (1.8, 6.0):
    report("  ... 40 refinement passes, b = %g" % b, partition(trw + b * ts + 40.0, passes=40))
