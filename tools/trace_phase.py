#!/usr/bin/env python3
"""Timeline of the backbone phases of the LAST step in a rocprofv3 --kernel-trace CSV of bench.py: every dispatch between
the optimizer's launch and the ROIAlign forward (backbone forward) and between the ROIAlign backward and the next
optimizer launch (backbone backward), in time order with its duration and the idle gap before it; then per-kernel sums.

    python3 tools/trace_phase.py /tmp/kt/r_kernel_trace.csv"""
import collections
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
from _trace_util import step_marks
opt = step_marks(rows)
fwd = [i for i, r in enumerate(rows) if ("roi_align_fwd" in r["Kernel_Name"] or "roi_align_wino7_pair" in r["Kernel_Name"])]
bwd = [i for i, r in enumerate(rows) if "roi_align_bwd" in r["Kernel_Name"] or "roi_partial_reduce" in r["Kernel_Name"]]
last_fwd = fwd[-1]
start = max(i for i in opt if i < last_fwd) + 1
phases = [("backbone forward", start, max(i for i in range(start, last_fwd) if "roi_tables" not in rows[i]["Kernel_Name"]) + 1),
          ("backbone backward", max(i for i in bwd if i > last_fwd) + 1, min([i for i in opt if i > last_fwd] + [len(rows)]))]
for title, a, b in phases:
    sel = rows[a:b]
    if not sel:
        continue
    t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
    print("== %s: %d dispatches, wall %.3f ms, kernel time %.3f ms, idle %.3f ms" % (title, len(sel), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    agg = collections.OrderedDict()
    prev = t0
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "-v" in sys.argv:
            print("  %8.1f us  gap %6.1f  %s" % ((e - s) / 1e3, (s - prev) / 1e3, name(r)))
        prev = max(prev, e)
        k = agg.setdefault(name(r), [0, 0])
        k[0] += 1
        k[1] += e - s
    for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("  %4d x  %8.1f us  %s" % (n, d / 1e3, k))
