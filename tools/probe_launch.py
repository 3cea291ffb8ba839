#!/usr/bin/env python
"""Fixed cost of a kernel launch in k_accumulate's shape (debug probe; needs an MI355X)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpose_amd import capi, synth  # noqa: E402

img, pts, tris, he, ratio = synth.workload(2048, 2048, 3000)
ctx = capi.Context(0, 2048, 2048)
ctx.set_image(capi.IMAGE_A, img)
ctx.upload(pts, tris, None)
ctx.iterate(capi.default_params(0), 16)
ctx.synchronize()
f = ctx.lib.tp_debug_null_launch
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
for name, mode, blocks, threads, lds in [
        ("empty 512x512, 66 KB LDS", 0, 512, 512, 67368), ("empty 512x512, no LDS", 0, 512, 512, 0),
        ("empty 512x64", 0, 512, 64, 0), ("empty 1x64", 0, 1, 64, 0), ("empty 153x256", 0, 153, 256, 0),
        ("8 MB of stores, 512x512", 1, 512, 512, 67368), ("16 MB read, 512x512, 66 KB LDS", 2, 512, 512, 67368),
        ("16 MB read, 2048x256", 2, 2048, 256, 0), ("LDS write+barrier, 512x512", 4, 512, 512, 67368),
        ("GRAPH empty 1x64", 16, 1, 64, 0), ("GRAPH empty 512x512 66 KB", 16, 512, 512, 67368),
        ("GRAPH empty 153x256", 16, 153, 256, 0),
        ("GRAPH 8 MB stores 512x512", 17, 512, 512, 67368), ("GRAPH 16 MB read 512x512", 18, 512, 512, 67368),
        ("GRAPH 16 MB read 2048x256", 18, 2048, 256, 0)]:
    us = C.c_double()
    for rep in range(2):
        rc = f(ctx.h, mode, blocks, threads, lds, 200, C.byref(us))
    print("%-40s rc=%d %.2f us" % (name, rc, us.value), flush=True)
