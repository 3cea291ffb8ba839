#!/usr/bin/env python
"""How the persistent kernel degrades with the input: microseconds per grad-iter at the metric size on the synthetic raster at contrasts
0.1 / 0.14 / 0.3 / 1.0 (1.0 = SURVEY section 8(d)'s raster as written), in calls of 20 grad-iters (the driver's shape) and of 2048, plus
overflow statistics.  python tools/contrast_sweep.py [W NT]"""
import sys, time
sys.path.insert(0, '.')
from tpose_amd import capi, synth
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
for contrast in (0.1, 0.14, 0.3, 1.0):
    img, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=contrast)
    c = capi.Context(0, W, H); c.set_image(capi.IMAGE_A, img); c.upload(pts, tris, None)
    p = capi.default_params(0); c.prepare(p); c.iterate(p, 5); c.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); c.iterate(p, 20); c.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
    short = sorted(ts)[2]
    out = []
    for steps in (256, 2048, 2048, 2048):
        t0 = time.perf_counter(); c.iterate(p, steps); c.synchronize(); out.append((time.perf_counter() - t0) / steps * 1e6)
    print("contrast %.2f: calls of 20: %.2f us/grad-iter | 256: %.2f | 2048: %.2f %.2f %.2f | replans %d given up %d"
          % (contrast, short, out[0], out[1], out[2], out[3], c.info(capi.INFO_REPLANS), c.info(capi.INFO_PERSIST_FAILURES)), flush=True)
    c.close()
