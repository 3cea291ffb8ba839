#!/usr/bin/env python
"""The persistent kernel at the metric size (2048^2 / 3000 triangles) on the reference's OWN pictures -- each resampled to 2048 x 2048 by the
integer resampler of tpose_amd/photos.py -- beside the bench's synthetic raster: microseconds per grad-iter in calls of 20 (the driver's
shape), of 2048, and of one 256-grad-iter launch between HIP events (the roofline's kernel figure), plus how far the mesh moves.
python tools/photo_timing.py [names ...]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from tpose_amd import capi, photos, synth
W = H = 2048
NT = 3000
names = sys.argv[1:] or ["synthetic_x0.10", "meninas", "fruit", "imageA", "shoeA"]
_, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=0.1)
B = 4 * W * H + 16 * NT + 24 * 13 * NT + 24 * pts.shape[0]
for name in names:
    if name.startswith("synthetic_x"):
        img = synth.workload(W, H, NT, contrast=float(name.split("x")[1]))[0]
    else:
        img = photos.resample_int(photos.load(name), W, H)
    c = capi.Context(0, W, H); c.set_image(capi.IMAGE_A, img); c.upload(pts, tris, None)
    p = capi.default_params(0); c.prepare(p); c.iterate(p, 5); c.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); c.iterate(p, 20); c.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
    short = sorted(ts)[2]
    c.iterate(p, 256 - 105); c.synchronize()
    out = []
    for steps in (2048, 2048, 2048):
        t0 = time.perf_counter(); c.iterate(p, steps); c.synchronize(); out.append((time.perf_counter() - t0) / steps * 1e6)
    ev = []
    for _ in range(5):
        c.timer_start(); c.iterate(p, 256); ev.append(c.timer_stop())
    k = sorted(ev)[2]
    moved = np.abs(c.retrieve(capi.BUF_POINTS) - pts).max(axis=1) * (H / 2)
    print("%-16s calls of 20: %.2f us/grad-iter | 2048: %.2f %.2f %.2f | one 256-grad-iter launch: %.1f us = %.3f us/grad-iter, frac %.3f | rows per lane %d | "
          "vertices moved (px, after %d grad-iters): median %.1f, 90 %% %.1f, max %.1f | replans %d (for balance %d; balance now %.2f, heaviest vertex %.2f) given up %d"
          % (name, short, out[0], out[1], out[2], k, k / 256, B * 256 / (k * 1e-6) / 8e12, c.info(13), 256 + 3 * 2048 + 5 * 256,
             np.median(moved), np.percentile(moved, 90), moved.max(), c.info(capi.INFO_REPLANS), c.info(14), c.info(15) / 1000.0, c.info(16) / 1000.0, c.info(capi.INFO_PERSIST_FAILURES)), flush=True)
    c.close()
