#!/usr/bin/env python3
"""Which kernels ran concurrently with a given kernel in the last step of a rocprofv3 --kernel-trace CSV.

    python3 tools/trace_overlap.py <kernel_trace.csv> sgd_multi_kernel
"""
import csv
import sys

path, pat = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if ("roi_align_fwd" in r["Kernel_Name"] or "roi_align_wino7_pair" in r["Kernel_Name"])]
sel = rows[marks[-1]:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    if pat not in r["Kernel_Name"]:
        continue
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%s  start %.3f ms  dur %.3f ms  queue %s" % (r["Kernel_Name"][:50], (a - t0) / 1e6, (b - a) / 1e6, r.get("Queue_Id")))
    for q in sel:
        if q is r:
            continue
        c, d = int(q["Start_Timestamp"]), int(q["End_Timestamp"])
        ov = min(b, d) - max(a, c)
        if ov > 0:
            print("    overlaps %-60s dur %.3f ms  overlap %.3f ms  queue %s" % (q["Kernel_Name"][:60], (d - c) / 1e6, ov / 1e6, q.get("Queue_Id")))
