cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -o r1 -- python bench.py --steps 512 --warmup 64 --no-cpu-baseline > gpurun_out/prof_r1.log 2>&1
tail -1 gpurun_out/prof_r1.log | cut -c1-300
f=$(find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -1); cat "$f" | head -8
cat > /tmp/t.py <<'PY'
import sys, time, os
sys.path.insert(0,'.')
import numpy as np
from tpose_amd import capi, synth
W=H=2048
img,pts,tris,he,ratio=synth.workload(W,H,3000)
ctx=capi.Context(0,W,H); ctx.set_image(0,img); ctx.upload(pts,tris)
p=capi.default_params(0)
us=ctx.profile_iterate(p,100)
print('DEBUG',os.environ.get('TPOSE_DEBUG_ACC','0'),'acc us %.2f'%us, 'visits', ctx.info(4))
PY
for d in 0 2; do TPOSE_DEBUG_ACC=$d python /tmp/t.py; done
