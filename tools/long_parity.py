import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
from oracle import oracle as O
from tpose_amd import capi, synth
from util import case, RATE
W, H = 300, 200
img, imgB, pts, tris, ratio, colors = case(W, H, (15, 5))
for flavour in (0, 1):
    ctx = capi.Context(0, W, H); ctx.set_image(capi.IMAGE_A, img); ctx.set_image(capi.IMAGE_B, imgB)
    ctx.upload(pts, tris, colors if flavour else None)
    p = capi.default_params(flavour)
    N = 20000
    t0 = time.time(); ctx.iterate(p, N); got = ctx.retrieve(capi.BUF_POINTS); tg = time.time() - t0
    t0 = time.time()
    ref = O.iterate(imgB if flavour else img, pts, tris, flavour, ratio, RATE[flavour], N, colors=colors if flavour else None, literal=False)
    to = time.time() - t0
    print("flavour", flavour, "iters", N, "points equal:", np.array_equal(got.view(np.uint32), ref["points"].view(np.uint32)),
          "ten equal:", np.array_equal(ctx.retrieve(capi.BUF_TENERGY), ref["ten"]), "gpu %.2fs oracle %.1fs" % (tg, to), flush=True)
    ctx.close()
