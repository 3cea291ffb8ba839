#!/usr/bin/env python
"""BASELINE config 2 on synthetic data: the full topology-optimisation schedule of software/triangulate up to
3000 triangles at the window size the reference would use for resource/meninas.png (1200x1381 / 1.5 = 800x920),
through the headless harness (piecewise API, four readbacks per frame, like the reference).  Needs an MI355X."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tpose_amd import synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tpose_amd", "host")
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
W, H = 1200, 1381   # meninas.png; the harness resamples to the reference's window (-window 1.5 -> 800x920)
img = synth.voronoi_raster(W, H, seed=1234, sites=160)
# The descent uses a fixed step, rate * dE / 65536 (shift.cs:45), and dE grows with raster area x contrast^2: on the
# full-contrast benchmark raster the steps are tens of pixels and the schedule's convergence test (relative energy
# change < 1e-4) is never met.  Photographs have far lower contrast between neighbouring regions; model that.
CONTRAST = float(os.environ.get("CONFIG2_CONTRAST", "0.1"))
rgb = img[:, :, :3].astype(np.float32)
img[:, :, :3] = np.clip(128.0 + (rgb - 128.0) * CONTRAST + 0.5, 0, 255).astype(np.uint8)
if os.environ.get("CONFIG2_PHOTO"):   # the picture config 2 names: resource/meninas.png (tests/golden/photos)
    from tpose_amd import photos
    img = photos.load(os.environ["CONFIG2_PHOTO"])
    H, W = img.shape[:2]
ppm = os.path.join(out, "config2.ppm")
with open(ppm, "wb") as f:
    f.write(b"P6\n%d %d\n255\n" % (W, H))
    f.write(np.ascontiguousarray(img[:, :, :3]).tobytes())
subprocess.check_call(["make", "-s", "-C", HOST, "triangulate"])
levels = "50,100,200,300,400,500,600,700,800,900,1000,1500,2000,2500,3000"
t0 = time.perf_counter()
extra = sys.argv[2:]  # e.g. -maxframes 20000
r = subprocess.run([os.path.join(HOST, "triangulate"), "-i", ppm, "-o", os.path.join(out, "config2.tri"), "-levels", levels,
                    "-window", "1.5"] + ([] if os.environ.get("CONFIG2_VERBOSE") else ["-quiet"]) + extra, capture_output=True, text=True, timeout=int(sys.argv[1]) if len(sys.argv) > 1 else 900)
dt = time.perf_counter() - t0
print("\n".join(r.stdout.strip().splitlines()[-(40 if os.environ.get("CONFIG2_VERBOSE") else 1):]))
for line in r.stderr.strip().splitlines()[-3:]:
    print(line)
print("wall incl. start-up %.1f s" % dt)
