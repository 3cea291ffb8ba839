import sys
sys.path.insert(0,'.')
import numpy as np
from tpose_amd import capi, synth
W,H=800,450
img = synth.photo_contrast(synth.voronoi_raster(W, H, seed=5, sites=12), 0.3)
imgB = synth.displaced_raster(img)
ratio = float(np.float32(W)/np.float32(H))
mode = sys.argv[1] if len(sys.argv) > 1 else "until"
ctxs = [capi.Context(0, W, H) for _ in range(2)]
for c in ctxs:
    c.set_image(capi.IMAGE_A, img); c.set_image(capi.IMAGE_B, imgB)
ctxs[1].set_persistent(False)
for grid in ((3,2),(5,5),(10,5),(14,7),(20,10),(3,2),(14,7)):
    pts, tris, _ = synth.grid_triangulation(grid[0], grid[1], ratio=ratio)
    colors = synth.mean_colors(img, pts, tris, ratio)
    out=[]
    for c in ctxs:
        c.upload(pts, tris, colors)
        p = capi.default_params(1)
        if mode == "until":
            tot = 1.0
            for rep in range(3):
                n, tot, _ = c.iterate_until(p, 60, 1e-6, tot)
        else:
            for n in (7, 60, 33):
                c.iterate(p, n)
        c.synchronize()
        out.append((c.retrieve(capi.BUF_POINTS), c.retrieve(capi.BUF_TENERGY)))
    same = np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32)) and np.array_equal(out[0][1], out[1][1])
    print(grid, "NT", tris.shape[0], "patches", ctxs[0].info(capi.INFO_PATCHES), "persist iters", ctxs[0].info(capi.INFO_PERSIST_ITERS), "same" if same else "DIFFERENT", flush=True)
