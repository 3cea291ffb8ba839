import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tpose_amd import capi, synth
img, pts, tris, he, ratio = synth.workload(1024, 1024, 3000)
free0 = torch.cuda.mem_get_info()[0]
for k in range(150):
    ctx = capi.Context(0, 1024, 1024)
    ctx.set_image(capi.IMAGE_A, img); ctx.set_image(capi.IMAGE_B, img)
    ctx.upload(pts, tris, synth.mean_colors(img, pts, tris, ratio) if k % 2 else None)
    ctx.iterate(capi.default_params(k % 2), 33)
    ctx.retrieve_many([capi.BUF_TENERGY, capi.BUF_POINTS])
    ctx.render(capi.RENDER_STORED if k % 2 else capi.RENDER_AVERAGE)
    ctx.close()
    if k in (10, 149):
        print(k, "free MB delta vs start:", (free0 - torch.cuda.mem_get_info()[0]) / 1e6, flush=True)
