#!/usr/bin/env python
"""In-kernel timeline of k_bin at the headline workload (TPOSE_DEBUG_ACC=16).  Needs an MI355X."""
import ctypes
import json
import os
import sys

os.environ["TPOSE_DEBUG_ACC"] = "16"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tpose_amd import capi, synth  # noqa: E402

img, pts, tris, he, ratio = synth.workload(2048, 2048, 3000)
ctx = capi.Context(0, 2048, 2048)
ctx.set_image(capi.IMAGE_A, img)
ctx.upload(pts, tris, None)
p = capi.default_params(capi.TRIANGULATE)
ctx.profile_iterate(p, 4)
ctx.iterate(p, 64)
ctx.synchronize()
lib = ctx.lib
lib.tp_debug_dump.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
acc = []
for rep in range(32):
    ctx.iterate(p, 1)
    ctx.synchronize()
    buf = np.zeros(512 * 16, np.uint64)
    assert lib.tp_debug_dump(ctx.h, buf.ctypes.data, buf.size) == 0
    acc.append(buf.reshape(512, 16).astype(np.int64))
st = np.stack(acc)
nb = int((st[0, :, 0] > 0).sum())
st = st[:, :nb]
t0 = st[:, :, 0].min(axis=1, keepdims=True)
names = ["start", "vertex stage done", "rectangles + scan done", "pair loop done", "end"]
out = {"blocks": nb}
for k, n in enumerate(names):
    v = (st[:, :, k] - t0) / 100.0
    out[n] = {"median": float(np.median(v)), "max": float(v.max(axis=1).mean())}
print(json.dumps(out, indent=1))
