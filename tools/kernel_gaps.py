#!/usr/bin/env python
"""Durations and gaps of k_lines / k_update from a `rocprofv3 --kernel-trace --output-format csv` directory, early and
late in the run (the descent changes the mesh).  usage: kernel_gaps.py <dir>"""
import csv
import glob
import statistics as st
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_lines", "k_update"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for name, part in (("first", rows[200:1200]), ("last", rows[-1000:])):
    d = {"k_lines": [], "k_update": []}
    for r in part:
        d[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    span = (int(part[-1]["End_Timestamp"]) - int(part[0]["Start_Timestamp"])) / (len(part) / 2)
    print(name, "k_lines median ns", st.median(d["k_lines"]), "k_update median ns", st.median(d["k_update"]), "ns per grad-iter", round(span))
