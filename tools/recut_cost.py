#!/usr/bin/env python
"""What a re-cut costs: per-grad-iter period of the workgroups around the grad-iters where a launch counts its lines' chunks again
(debug flavour; TPOSE_DBG_FIRST=40 stamps grad-iters 40..103 of a launch, the re-cut is at 64)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TPOSE_HIP_LIB"] = os.path.join(ROOT, "tpose_amd", "variants", "libtpose_hip_debug.so")
os.environ["TPOSE_DBG_FIRST"] = "40"
import numpy as np
from tpose_amd import capi, synth
img, pts, tris, he, ratio = synth.workload(2048, 2048, 3000, contrast=0.1)
ctx = capi.Context(0, 2048, 2048); ctx.set_image(capi.IMAGE_A, img); ctx.upload(pts, tris, None)
p = capi.default_params(0); ctx.iterate(p, 600); ctx.iterate(p, 200); ctx.synchronize()
lib = ctx.lib; lib.tp_debug_dump_persist.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(512 * 64 * 16, np.uint64); assert lib.tp_debug_dump_persist(ctx.h, buf.ctypes.data, buf.size) == 0
st = buf.reshape(512, 64, 16)[:256].astype(np.int64)
per = np.diff(st[:, :, 0], axis=1) / 100.0   # [wg, it] period starting at it
med = np.median(per, axis=0)
print("median period of grad-iters 40+k: ", " ".join("%d:%.1f" % (40 + k, med[k]) for k in range(16, 36)))
print("sum over 60..72 minus 13 x steady %.2f = %.1f us extra around the re-cut" % (np.median(med[:14]), med[20:33].sum() - 13 * np.median(med[:14])))
for k, lab in enumerate(["P0", "P1+recut", "P3", "P6"]):
    d = (st[:, :, k + 1] - st[:, :, k]) / 100.0
    print(lab, "steady %.2f, at the re-cut grad-iter %.2f, the one after %.2f" % (np.median(d[:, 5:20]), np.median(d[:, 24]), np.median(d[:, 25])))
