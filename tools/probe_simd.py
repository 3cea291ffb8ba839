#!/usr/bin/env python
"""Which SIMD / CU / XCC the eight waves of k_accumulate-shaped workgroups land on (debug probe, MI355X)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tpose_amd import capi, synth  # noqa: E402

img, pts, tris, he, ratio = synth.workload(2048, 2048, 3000)
ctx = capi.Context(0, 2048, 2048)
ctx.set_image(capi.IMAGE_A, img)
ctx.upload(pts, tris, None)
ctx.iterate(capi.default_params(0), 16)
ctx.synchronize()
f = ctx.lib.tp_debug_null_launch
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
us = C.c_double()
assert f(ctx.h, 5, 768, 512, 50856, 1, C.byref(us)) == 0
g = ctx.lib.tp_debug_read_visits
g.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
buf = np.zeros(768 * 16 * 2, np.uint32)
assert g(ctx.h, buf.ctypes.data, buf.nbytes) == 0
hw = buf.reshape(768, 16, 2)[:, :8, :]
simd = (hw[:, :, 0] >> 4) & 3
cu = (hw[:, :, 0] >> 8) & 15
sh = (hw[:, :, 0] >> 12) & 1
se = (hw[:, :, 0] >> 13) & 7
xcc = hw[:, :, 1] & 15
for b in (0, 1, 2, 8, 256, 257, 512, 513, 767):
    print("wg %3d: simd per wave %s  cu %d sh %d se %d xcc %d" % (b, simd[b].tolist(), cu[b, 0], sh[b, 0], se[b, 0], xcc[b, 0]))
pat = {}
for b in range(768):
    pat[tuple(simd[b].tolist())] = pat.get(tuple(simd[b].tolist()), 0) + 1
print("simd patterns:", sorted(pat.items(), key=lambda kv: -kv[1])[:8])
key = (xcc[:, 0].astype(np.int64) << 16) | (se[:, 0].astype(np.int64) << 12) | (sh[:, 0].astype(np.int64) << 8) | cu[:, 0]
uniq, counts = np.unique(key, return_counts=True)
print("distinct CUs used:", len(uniq), "workgroups per CU histogram:", np.bincount(counts).tolist())
# which workgroups share a CU with workgroup 0?
print("workgroups on the CU of wg 0:", np.where(key == key[0])[0].tolist())
