import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tpose_amd import capi, synth
img, pts, tris, he, ratio = synth.workload(2048, 2048, 3000)
for name, mut in [("nan", lambda p: np.where(np.arange(p.shape[0])[:, None] % 7 == 5, np.nan, p).astype(np.float32)),
                  ("inf", lambda p: np.where(np.arange(p.shape[0])[:, None] % 7 == 5, np.inf, p).astype(np.float32)),
                  ("huge", lambda p: (p * 1e6).astype(np.float32)),
                  ("allsame", lambda p: np.zeros_like(p))]:
    ctx = capi.Context(0, 2048, 2048)
    ctx.set_image(capi.IMAGE_A, img)
    q = mut(pts.copy())
    ctx.upload(q, tris, None)
    p = capi.default_params(0)
    t0 = time.perf_counter()
    try:
        ctx.iterate(p, 4)
        ctx.synchronize()
        msg = "ok"
    except Exception as e:
        msg = "error: " + str(e)[:100]
    print(name, "%.1f ms for 4 iters" % ((time.perf_counter() - t0) * 1e3), msg, flush=True)
    ctx.close()
