import sys, time
sys.path.insert(0, '.')
from tpose_amd import capi, synth
img, pts, tris, he, ratio = synth.workload(2048, 2048, 3000, contrast=0.1)
c = capi.Context(0, 2048, 2048); c.set_image(capi.IMAGE_A, img); c.upload(pts, tris, None)
p = capi.default_params(0); c.prepare(p)
out=[]
for k in range(24):
    c.timer_start(); c.iterate(p, 256); out.append(c.timer_stop()/256)
print("us per grad-iter by launch of 256:", " ".join("%.2f"%v for v in out), "| replans", c.info(capi.INFO_REPLANS), "warm", c.info(capi.INFO_WARM_LAUNCHES))
