"""20 000 grad-iters of the persistent kernel against the oracle, bit for bit -- in one call, and in calls of odd
lengths (1, 2, 3, 63, 65, 511, 513, 777 ... grad-iters: every launch ends and restarts the position hand-over).
A stale cross-workgroup read shows up here.  Needs an MI355X:  python tools/long_parity.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
sys.path.insert(0, "tests")
import numpy as np

from oracle import oracle as O
from tpose_amd import capi
from util import RATE, case

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
W, H = 300, 200
img, imgB, pts, tris, ratio, colors = case(W, H, (15, 5))
ok = True
for flavour in (0, 1):
    t0 = time.time()
    ref = O.iterate(imgB if flavour else img, pts, tris, flavour, ratio, RATE[flavour], N, colors=colors if flavour else None, literal=False)
    to = time.time() - t0
    for mode in ("one call", "odd calls"):
        ctx = capi.Context(0, W, H)
        ctx.set_image(capi.IMAGE_A, img)
        ctx.set_image(capi.IMAGE_B, imgB)
        ctx.upload(pts, tris, colors if flavour else None)
        p = capi.default_params(flavour)
        t0 = time.time()
        if mode == "one call":
            ctx.iterate(p, N)
        else:
            left, k = N, 0
            sizes = [1, 2, 3, 63, 65, 511, 513, 777, 5, 1025, 129]
            while left > 0:
                n = min(left, sizes[k % len(sizes)])
                ctx.iterate(p, n)
                left -= n
                k += 1
        got = ctx.retrieve(capi.BUF_POINTS)
        tg = time.time() - t0
        e = (np.array_equal(got.view(np.uint32), ref["points"].view(np.uint32)), np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"]),
             np.array_equal(ctx.retrieve(capi.BUF_GRADIENT), ref["gr"]))
        ok &= all(e)
        print("flavour %d, %d grad-iters, %s: points %s tenergy %s gradient %s | persistent grad-iters %d in %d launches, patches %d, "
              "give-ups %d | gpu %.2f s, oracle %.1f s" % (flavour, N, mode, e[0], e[1], e[2], ctx.info(capi.INFO_PERSIST_ITERS),
                                                         ctx.info(capi.INFO_PERSIST_LAUNCHES), ctx.info(capi.INFO_PATCHES), ctx.info(9), tg, to), flush=True)
        ctx.close()
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
