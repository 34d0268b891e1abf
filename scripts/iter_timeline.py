#!/usr/bin/env python3
"""Dev tool: the kernels of one LM iteration from a rocprofv3 kernel trace (start offset, duration, gap to the previous kernel).
usage: iter_timeline.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_lm_control" in r["Kernel_Name"]]
i0, i1 = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0, prev, n = int(rows[i0]["Start_Timestamp"]), None, 0
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    if "k_ldlt_step" in name:
        n += 1
        if 2 < n < 36:
            continue
    print("%8.1f  %-34s dur %6.1f gap %5.1f" % ((s - t0) / 1e3, name, (e - s) / 1e3, gap))
print("iteration %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
