cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, time
sys.path.insert(0,'.')
import numpy as np
from tpose_amd import capi, synth
W=H=2048
img,pts,tris,he,ratio=synth.workload(W,H,3000)
ctx=capi.Context(0,W,H); ctx.set_image(0,img); ctx.upload(pts,tris)
p=capi.default_params(0)
prev=pts.copy()
for it in range(1,2001):
    ctx.iterate(p,1)
    if it in (1,2,3,5,10,20,50,100,200,500,1000,2000):
        cur=ctx.retrieve(capi.BUF_POINTS)
        ten=ctx.retrieve(capi.BUF_TENERGY)
        print(it,'max step px %.3f'%(np.abs(cur-prev).max()*1024),'mean %.4f'%(np.abs(cur-prev).mean()*1024),'E %.4e'%ten[:3000].astype(np.int64).sum())
    if it+1 in (1,2,3,5,10,20,50,100,200,500,1000,2000): prev=ctx.retrieve(capi.BUF_POINTS)
for m in (6,16,32,64):
    ctx=capi.Context(0,W,H); ctx.set_margin(m); ctx.set_image(0,img); ctx.upload(pts,tris)
    ctx.iterate(p,256); ctx.synchronize(); r0=ctx.info(6)
    t0=time.time(); ctx.iterate(p,2048); ctx.synchronize(); dt=time.time()-t0
    print('margin',m,'rebuilds first256',r0,'next2048',ctx.info(6)-r0,'pairs',ctx.info(4),'us/iter %.2f'%(dt/2048*1e6))
PY
