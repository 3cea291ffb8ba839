cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
cat > /tmp/t.py <<'PY'
import sys, time, os
sys.path.insert(0,'.')
import numpy as np
from tpose_amd import capi, synth
W=H=2048
img,pts,tris,he,ratio=synth.workload(W,H,3000)
ctx=capi.Context(0,W,H); ctx.set_image(0,img); ctx.upload(pts,tris)
p=capi.default_params(0)
ctx.iterate(p,40); ctx.synchronize()
PY
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc -o p1 -- python /tmp/t.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc -o p2 -- python /tmp/t.py > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'][:14]; acc[k][row['Counter_Name']]+=float(row['Counter_Value'])
    for k,v in acc.items():
        if k.startswith('k_'): print(k, {c: round(x/40) for c,x in v.items()})
PY
