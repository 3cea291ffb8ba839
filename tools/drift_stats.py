"""How a plan of the persistent kernel ages while the mesh drifts (CPU replay, tests/emul): per grad-iter, the lanes
that re-fetch a record, and the rows that no longer fit the records a lane keeps.  python tools/drift_stats.py [iters]"""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

from tpose_amd import synth

HERE = os.path.join(os.getcwd(), "tests")
so = os.path.join(HERE, "_build", "libtp_emul_persist_stats.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "emul", "emul_persist.cpp")])
emp = C.CDLL(so)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
W = H = 2048
img, pts, tris, he, ratio = synth.workload(W, H, 3000, contrast=0.1)
ws = np.zeros((iters, 6), np.int64)
recuts = np.zeros(1, np.int64)
emp.emul_persist_walk_stats(ws.ctypes.data_as(C.c_void_p), iters, recuts.ctypes.data_as(C.c_void_p))
p = np.ascontiguousarray(pts, np.float32).copy()
t4 = np.ascontiguousarray(tris, np.int32)
stats = np.zeros(16, np.int64)
start = p.copy()
sys.path.insert(0, HERE)
from oracle import oracle as O
d = O.dp(0, t4.shape[0])
rc = emp.emul_persist(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.strides[0]), W, H, p.ctypes.data_as(C.c_void_p), p.shape[0],
                      t4.ctypes.data_as(C.c_void_p), t4.shape[0], None, 0, C.c_float(d), C.c_float(ratio), C.c_float(0.00005), iters, 256,
                      160 * 1024 - 512, stats.ctypes.data_as(C.c_void_p), None, None, None, None)
print("rc", rc, "patches", stats[1], "lines", stats[3], "dp", d, "dp_px", d * 0.5 * H)
mv = np.abs((p - start) * np.array([W / 2, H / 2])).max(axis=1)
print("moved after %d grad-iters (px): median %.2f p90 %.2f max %.2f" % (iters, np.median(mv), np.percentile(mv, 90), mv.max()))
print("patch re-cuts that changed something:", recuts[0])
print("iter | stale lanes | most lane-items in a patch | lanes over | rows over | most rows over in a patch | most stale lanes in a patch")
for it in list(range(0, 8)) + list(range(8, iters, max(1, iters // 24))):
    print(it, *ws[it])
