#!/usr/bin/env python
"""Long descents in calls of mixed lengths, persistent path against the two-kernel path (bit for bit: positions, energies, gradient):
2048^2 / 3000 triangulate 120 000 grad-iters, warp 60 000, 4096^2 / 12 000 12 000.  Needs an MI355X.  python tools/long_mixed_calls.py"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tpose_amd import capi, synth
for (W, NT, flavour, iters, calls) in ((2048, 3000, 0, 120000, (20, 257, 2048, 5000, 33)), (2048, 3000, 1, 60000, (64, 1000, 7)), (4096, 12000, 0, 12000, (512, 100, 3000))):
    img, pts, tris, he, ratio = synth.workload(W, W, NT, contrast=0.1)
    if os.environ.get("TPOSE_PHOTO") and W == 2048:   # (round 6: the same on a reference photograph -- hot patches, 24 rows per lane, plans cut for balance)
        from tpose_amd import photos
        img = photos.resample_int(photos.load(os.environ["TPOSE_PHOTO"]), W, W)
    imgB = synth.displaced_raster(img); colors = synth.mean_colors(img, pts, tris, ratio) if flavour else None
    out = []
    for persistent in (1, 0):
        c = capi.Context(0, W, W); c.set_image(capi.IMAGE_A, img); c.set_image(capi.IMAGE_B, imgB); c.upload(pts, tris, colors)
        c.set_persistent(persistent)
        p = capi.default_params(flavour)
        done, k = 0, 0
        t0 = time.perf_counter()
        while done < iters:
            n = min(calls[k % len(calls)], iters - done) if persistent else min(4096, iters - done)
            c.iterate(p, n); done += n; k += 1
        c.synchronize(); dt = time.perf_counter() - t0
        out.append((c.retrieve(capi.BUF_POINTS), c.retrieve(capi.BUF_TENERGY), c.retrieve(capi.BUF_GRADIENT), dt, c.info(capi.INFO_PERSIST_ITERS), c.info(capi.INFO_PERSIST_FAILURES), c.info(capi.INFO_REPLANS)))
        c.close()
    same = all(np.array_equal(out[0][i].view(np.uint32), out[1][i].view(np.uint32)) for i in range(3))
    print(("photo %s, " % os.environ["TPOSE_PHOTO"] if os.environ.get("TPOSE_PHOTO") and W == 2048 else "") + "%d^2 / %d flavour %d, %d grad-iters in calls of %s: persistent %.2f us/iter (%d inside launches, %d given up, %d plans cut again) | two-kernel %.2f us/iter | positions, energies, gradient %s"
          % (W, NT, flavour, iters, calls, out[0][3] / iters * 1e6, out[0][4], out[0][5], out[0][6], out[1][3] / iters * 1e6, "the same" if same else "DIFFERENT"), flush=True)
