#!/usr/bin/env python
"""Grad-iter time for COARSE meshes on a large raster (the early states of the triangulate schedule,
software/triangulate/main.cpp:206-351, which starts from two triangles).  Needs an MI355X."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpose_amd import capi, synth  # noqa: E402

W = H = 2048
img = synth.voronoi_raster(W, H, seed=1234)
for gx, gy in [(1, 1), (2, 2), (4, 4), (8, 8), (16, 16), (32, 24)]:
    pts, tris, he = synth.grid_triangulation(gx, gy, ratio=1.0)
    ctx = capi.Context(0, W, H)
    ctx.set_image(capi.IMAGE_A, img)
    ctx.upload(pts, tris, None)
    p = capi.default_params(capi.TRIANGULATE)
    ctx.iterate(p, 64)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.iterate(p, 512)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(raster=[W, H], triangles=int(tris.shape[0]), us_per_iter=dt / 512 * 1e6,
                          chunks_per_line=ctx.info(1))), flush=True)
    ctx.close()
