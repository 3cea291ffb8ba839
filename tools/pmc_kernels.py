#!/usr/bin/env python
"""Per-kernel hardware counters of the headline workload: runs `rocprofv3 --pmc <counters>` (counters only, no trace
domains; one pass per group) over tools/time_acc.py and prints the per-launch average of every counter for every
kernel.  Needs an MI355X.   python tools/pmc_kernels.py [out.json]
TPOSE_PMC_TARGET=persist: over `bench.py --pmc-child` instead (16 persistent launches of 256 grad-iters on the bench's raster; the
census launch -- the same kernel, a few microseconds -- is left out of k_persist's averages)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"],
    ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
    ["GRBM_GUI_ACTIVE", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_THREAD_CYCLES_VALU", "SQ_INST_CYCLES_SALU", "SQ_WAIT_INST_ANY"],
    ["FETCH_SIZE"], ["WRITE_SIZE"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"], ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"],
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TOTAL_ACCESSES_sum"],
    ["TCP_TA_TCP_STATE_READ_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_GATE_EN1_sum"],
    ["TA_BUSY_avr", "TA_FLAT_READ_WAVEFRONTS_sum"],
    ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_REQUEST_sum"],
]
if os.environ.get("TPOSE_PMC_GROUPS"):  # e.g. "5,6,7": only these groups
    GROUPS = [GROUPS[int(k)] for k in os.environ["TPOSE_PMC_GROUPS"].split(",")]
exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
out = {}
for grp in GROUPS:
    d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    env = dict(os.environ, TPOSE_TIME_ACC_SHORT="1")
    target = [os.path.join(ROOT, "bench.py"), "--pmc-child"] if os.environ.get("TPOSE_PMC_TARGET") == "persist" else [os.path.join(ROOT, "tools", "time_acc.py")]
    cmd = [exe, "--pmc"] + grp + ["--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable] + target
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=d, env=env, timeout=150)
    except subprocess.TimeoutExpired:
        print("group timed out:", grp, file=sys.stderr)
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        print("group failed:", grp, r.returncode, r.stderr[-400:], file=sys.stderr)
        continue
    vals = {}
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        vals.setdefault((k, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    # (dispatch order is the same for every counter: the census of resident workgroups and tp_prepare's 8-grad-iter probe are the first two
    # launches of k_persist -- the same kernel, a few microseconds / 8 grad-iters -- and are left out of its averages)
    for (k, c), v in vals.items():
        if k == "k_persist" and len(v) > 2 and os.environ.get("TPOSE_PMC_TARGET") == "persist":
            v = v[2:]
        out.setdefault(k, {})[c] = round(sum(v) / len(v), 1)
        out[k]["launches"] = len(v)
    shutil.rmtree(d, ignore_errors=True)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(json.dumps(out, indent=1, sort_keys=True))
txt = json.dumps(out, indent=1, sort_keys=True)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt)
