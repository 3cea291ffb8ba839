# two contexts on one GPU iterating concurrently from two threads: persistent launches collide; results must still be exact
import sys, os, threading
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
from oracle import oracle as O
from tpose_amd import capi, synth
W, H = 640, 480
img = synth.photo_contrast(synth.voronoi_raster(W, H, seed=7, sites=12), 0.3)
ratio = float(np.float32(W) / np.float32(H))
pts, tris, _ = synth.grid_triangulation(50, 30, ratio=ratio)
N = 40
ref = O.iterate(img, pts, tris, 0, ratio, 0.00005, N, literal=False)
out = {}
def work(k):
    ctx = capi.Context(0, W, H); ctx.set_image(capi.IMAGE_A, img); ctx.upload(pts, tris)
    p = capi.default_params(0)
    for rep in range(10):
        ctx.upload(pts, tris)
        ctx.iterate(p, N)
        got = ctx.retrieve(capi.BUF_POINTS); ten = ctx.retrieve(capi.BUF_TENERGY)
        ok = np.array_equal(got.view(np.uint32), ref["points"].view(np.uint32)) and np.array_equal(ten, ref["ten"])
        out.setdefault(k, []).append(ok)
    out[(k, 'info')] = (ctx.info(capi.INFO_CENSUS), ctx.info(9), ctx.info(capi.INFO_PERSIST_ITERS))
    ctx.close()
ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
print(out)
sys.exit(0 if all(all(v) for k, v in out.items() if not isinstance(k, tuple)) else 1)
