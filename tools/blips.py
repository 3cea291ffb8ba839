#!/usr/bin/env python
"""Looking for slow stretches: calls of 512 grad-iters at the metric size, timed one by one on the host (each waited for), for N calls; prints
the outliers with the plan's state around them.  python tools/blips.py [N]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from tpose_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
img, pts, tris, he, ratio = synth.workload(2048, 2048, 3000, contrast=0.1)
c = capi.Context(0, 2048, 2048); c.set_image(capi.IMAGE_A, img); c.upload(pts, tris, None)
p = capi.default_params(0); c.prepare(p); c.iterate(p, 64); c.synchronize()
rows = []
for k in range(N):
    t0 = time.perf_counter(); c.iterate(p, 512); t1 = time.perf_counter(); c.synchronize(); t2 = time.perf_counter()
    rows.append(((t2 - t0) / 512 * 1e6, (t1 - t0) * 1e6, c.info(capi.INFO_REPLANS), c.info(13), c.info(capi.INFO_PERSIST_FAILURES), c.info(capi.INFO_WARM_LAUNCHES)))
us = np.array([r[0] for r in rows])
med = float(np.median(us))
print("median %.2f us per grad-iter over %d calls of 512; min %.2f max %.2f" % (med, N, us.min(), us.max()))
for k, r in enumerate(rows):
    flag = " <--" if r[0] > med * 1.08 else ""
    print("%3d: %.2f us/iter | tp_iterate returned after %.0f us | replans %d rows per lane %d given up %d warm %d%s" % ((k,) + r + (flag,)))
