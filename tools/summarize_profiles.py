#!/usr/bin/env python
"""gpurun_out/final/ (tools/collect_profiles.sh) -> profiles/rNN_* ; usage: summarize_profiles.py r01"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst, tag = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles"), sys.argv[1]


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
stats = find("kt/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
trace = find("kt/**/*kernel_trace.csv")
if trace:  # split the graph-replayed and the eager (event-bracketed) populations: the eager ones are the last 256 iterations
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    out = {}
    for name in ("k_accumulate", "k_bin", "k_reduce", "k_update"):
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if name + "(" in r["Kernel_Name"] or name + "<" in r["Kernel_Name"]]
        eager, graph = d[-256:], d[:-256]
        out[name] = {"graph_replay_avg_us": sum(graph) / max(len(graph), 1) / 1e3, "graph_replay_launches": len(graph),
                     "eager_avg_us": sum(eager) / max(len(eager), 1) / 1e3, "eager_launches": len(eager)}
    json.dump(out, open(os.path.join(dst, tag + "_kernel_split.json"), "w"), indent=1)
pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = find("pmc_%s/**/*counter_collection.csv" % c)
    if not f:
        continue
    acc = {}
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != c:
            continue
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
        s = acc.setdefault(k, [0.0, 0])
        s[0] += float(r["Counter_Value"]); s[1] += 1
    for k, (tot, n) in acc.items():
        if k.startswith("k_"):
            pmc.setdefault(k, {})[c + "_KB_per_launch"] = round(tot / n, 1)
for k, v in pmc.items():
    if "FETCH_SIZE_KB_per_launch" in v and "WRITE_SIZE_KB_per_launch" in v:
        v["hbm_bytes_per_launch_corrected"] = int((2 * v["FETCH_SIZE_KB_per_launch"] + v["WRITE_SIZE_KB_per_launch"]) * 1024)
if pmc:
    pmc["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (bench.py --steps 64 --warmup 16, 2048^2/3000); "
                    "units KB; gfx950 correction: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated")
    json.dump(pmc, open(os.path.join(dst, tag + "_pmc_hbm.json"), "w"), indent=1)
for name, to in (("configs.jsonl", "_configs.jsonl"), ("acc_timeline.json", "_accumulate_timeline.json"), ("coarse.jsonl", "_coarse_meshes.jsonl"),
                 ("launch_probe.txt", "_launch_probe.txt"), ("config2.txt", "_config2_schedule.txt"), ("config3.txt", "_config3_warp.txt"),
                 ("config4.json", "_config4_batch.json")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, tag + to))
print("profiles updated:", sorted(f for f in os.listdir(dst) if f.startswith(tag)))
