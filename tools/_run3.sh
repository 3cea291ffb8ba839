python tools/launch_profile.py 2>&1 | tail -30
python tools/persist_timeline.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a in ('1','50','100','mean','P0','P1','P3','P6','P7')})
    else: print(k, v)"
