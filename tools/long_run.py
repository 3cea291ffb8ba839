#!/usr/bin/env python
"""How the figure ages: microseconds per grad-iter of calls of 4096 at the metric size, every 16384 grad-iters up to N (default 131072).
python tools/long_run.py [N]"""
import sys, time
sys.path.insert(0, '.')
from tpose_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
img, pts, tris, he, ratio = synth.workload(2048, 2048, 3000, contrast=0.1)
c = capi.Context(0, 2048, 2048); c.set_image(capi.IMAGE_A, img); c.upload(pts, tris, None)
p = capi.default_params(0); c.prepare(p)
done = 0
while done < N:
    t0 = time.perf_counter()
    for _ in range(4):
        c.iterate(p, 4096)
    c.synchronize(); dt = time.perf_counter() - t0
    done += 16384
    print("%7d grad-iters: %.2f us per grad-iter | plans cut again %d, given up %d" % (done, dt / 16384 * 1e6, c.info(capi.INFO_REPLANS), c.info(capi.INFO_PERSIST_FAILURES)), flush=True)
