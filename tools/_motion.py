import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tpose_amd import capi, synth
W=H=2048; NT=3000
img, pts, tris, he, ratio = synth.workload(W,H,NT)
ctx = capi.Context(0,W,H); ctx.set_image(capi.IMAGE_A,img); ctx.upload(pts,tris,None)
p = capi.default_params(0)
prev = pts.copy()
out=[]
for it in range(1, 1201):
    ctx.iterate(p,1)
    if it in (1,2,3,5,10,20,50,100,200,300,500,800,1200) or it%100==0:
        cur = ctx.retrieve(capi.BUF_POINTS).reshape(-1,2)
        d = np.abs(cur-prev)*np.array([W/2/ratio, H/2])
        out.append((it, float(d.max()), float(np.percentile(d.max(axis=1),99)), float(np.median(d.max(axis=1)))))
    if True:
        prev = ctx.retrieve(capi.BUF_POINTS).reshape(-1,2).copy() if (it+1) in (1,2,3,5,10,20,50,100,200,300,500,800,1200) or (it+1)%100==0 else prev
for o in out: print("iter %d: per-iteration displacement px max %.2f p99 %.2f median %.3f" % o)
