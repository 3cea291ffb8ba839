#!/usr/bin/env python
"""Kernel-level timing of the headline workload (2048^2 / 3000 triangles) for A/B builds of the library:
TPOSE_HIP_LIB=<variant .so> python tools/time_acc.py [W H gx gy] -> one JSON line with the k_lines time
(graph replays of back-to-back launches, HIP events) and the fused grad-iter time.  Needs an MI355X."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("TPOSE_TIME_ACC_TORCH"):  # rocprofv3 --kernel-trace only survives this script with torch's runtime loaded first
    import torch  # noqa: F401
from tpose_amd import capi, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
NT = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
flavour = int(sys.argv[4]) if len(sys.argv) > 4 else 0
pre = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # grad-iters before anything is timed (the descent degrades the mesh)
img, pts, tris, he, ratio = synth.workload(W, H, NT)
ctx = capi.Context(0, W, H)
ctx.set_image(capi.IMAGE_A, img)
colors = None
if flavour == 1:
    ctx.set_image(capi.IMAGE_B, synth.displaced_raster(img))
    colors = synth.mean_colors(img, pts, tris, ratio)
ctx.upload(pts, tris, colors)
p = capi.default_params(flavour)
if os.environ.get("TPOSE_TIME_ACC_RATE0"):  # timing experiments whose sums are wrong: nothing moves
    p.rate = 0.0
ctx.prepare(p)
ctx.iterate(p, 64 + pre)
ctx.synchronize()
if os.environ.get("TPOSE_TIME_ACC_REUPLOAD"):  # the degraded mesh uploaded afresh: the chunk count per line is chosen again
    ctx.upload(ctx.retrieve(capi.BUF_POINTS).reshape(-1, 2), tris, colors)
    ctx.prepare(p)
    ctx.iterate(p, 64)
    ctx.synchronize()
if os.environ.get("TPOSE_TIME_ACC_SHORT"):  # under a counter pass: a few dozen launches are enough
    ctx.iterate(p, 32)
    ctx.synchronize()
    ctx.close()
    sys.exit(0)
acc = sorted(ctx.profile_accumulate(p, 64) for _ in range(5))
its = []
for _ in range(5):
    t0 = time.perf_counter()
    ctx.iterate(p, 1024)
    ctx.synchronize()
    its.append((time.perf_counter() - t0) / 1024 * 1e6)
its.sort()
print(json.dumps({"lib": os.path.basename(capi.LIB_PATH), "raster": [W, H], "triangles": tris.shape[0], "flavour": flavour, "pre_iters": pre,
                  "k_lines_us": round(acc[2], 3), "k_lines_us_min": round(acc[0], 3),
                  "iter_us": round(its[2], 3), "iter_us_min": round(its[0], 3),
                  "prefix_pitch": ctx.info(0), "chunks_per_line": ctx.info(1)}), flush=True)
ctx.close()
