#!/usr/bin/env python
"""What the driver's call (--warmup 5, then ONE call of 20 grad-iters) is made of: wall clock of the call, the same between HIP events on the
library's stream, and the host's part alone (tp_iterate returning, before the synchronize).  python tools/call_parts.py [names ...]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from tpose_amd import capi, photos, synth
W = H = 2048; NT = 3000
_, pts, tris, he, ratio = synth.workload(W, H, NT, contrast=0.1)
for name in sys.argv[1:] or ["meninas", "synthetic_x0.10"]:
    img = synth.workload(W, H, NT, contrast=float(name.split("x")[1]))[0] if name.startswith("synthetic_x") else photos.resample_int(photos.load(name), W, H)
    rows = []
    for rep in range(5):
        c = capi.Context(0, W, H); c.set_image(capi.IMAGE_A, img); c.upload(pts, tris, None)
        p = capi.default_params(0); c.prepare(p); c.iterate(p, 5); c.synchronize()
        use_events = rep % 2 == 1
        if use_events: c.timer_start()
        t0 = time.perf_counter(); c.iterate(p, 20); t1 = time.perf_counter()
        ev = c.timer_stop() if use_events else float("nan")
        c.synchronize(); t2 = time.perf_counter()
        rows.append(((t2 - t0) * 1e6, (t1 - t0) * 1e6, ev))
        c.close()
    print("%-16s call of 20 behind prepare + 5: wall %s us | tp_iterate returns after %s | between HIP events (alternate runs) %s"
          % (name, " ".join("%.0f" % r[0] for r in rows), " ".join("%.0f" % r[1] for r in rows), " ".join("%.0f" % r[2] for r in rows)), flush=True)
