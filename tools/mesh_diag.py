import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from tpose_amd import capi, synth
W=H=2048
contrast=float(sys.argv[1]) if len(sys.argv)>1 else 1.0
img, pts, tris, he, ratio = synth.workload(W,H,3000,contrast=contrast)
print('contrast',contrast)
ctx = capi.Context(0,W,H); ctx.set_image(capi.IMAGE_A, img); ctx.upload(pts,tris)
ctx.set_persistent(0)
p = capi.default_params(0)
e = set()
for t in tris:
    for k in range(3):
        a,b = int(t[k]), int(t[(k+1)%3]); e.add((min(a,b),max(a,b)))
e = np.array(sorted(e))
done=0
for n in [0,64,1000,4000,16000]:
    ctx.iterate(p, n-done) if n>done else None; done=n
    q = ctx.retrieve(capi.BUF_POINTS)
    d = q[e[:,0]]-q[e[:,1]]
    rows = np.abs(d[:,1])*H/2; cols=np.abs(d[:,0])*W/2/ratio
    mv = np.abs(q-pts).max()*H/2
    print(n, "rows mean %.1f p99 %.1f max %.1f | cols max %.1f | max vertex displacement %.1f px | total rows %.0f" % (rows.mean(), np.percentile(rows,99), rows.max(), cols.max(), mv, rows.sum()))
