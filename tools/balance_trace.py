#!/usr/bin/env python
"""How the plans follow the mesh: the metric workload on a picture (TPOSE_PHOTO, default meninas), chunk by chunk of 512 grad-iters -- us per
grad-iter, plans cut so far, the model's balance of the current plan."""
import os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("TPOSE_PHOTO", "meninas")
from tpose_amd import capi, photos
img, pts, tris, he, ratio, label = photos.raster_from_env(2048, 2048, 3000)
c = capi.Context(0, 2048, 2048); c.set_image(capi.IMAGE_A, img); c.upload(pts, tris, None)
p = capi.default_params(0); c.prepare(p)
print(label)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    t0 = time.perf_counter(); c.iterate(p, 512); c.synchronize(); dt = (time.perf_counter() - t0) / 512 * 1e6
    print("chunk %2d: %.2f us/grad-iter | replans %d (balance %d) | model balance %.2f heaviest vertex %.2f | rows per lane %d"
          % (k, dt, c.info(8), c.info(14), c.info(15) / 1e3, c.info(16) / 1e3, c.info(13)), flush=True)
