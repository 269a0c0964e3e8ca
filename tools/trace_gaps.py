#!/usr/bin/env python3
"""GPU idle gaps of the last step of a rocprofv3 --kernel-trace CSV: the largest intervals in which no kernel runs on
any queue, with the kernels on either side.

    python3 tools/trace_gaps.py <kernel_trace.csv> [top]

Caveat: the profiler slows the HOST (a traced step takes ~2 ms longer), so gaps that only exist because the host fell behind
(e.g. before the optimizer launch at the end of backward) are overstated; without the profiler the host queues the whole
backward while the GPU is inside the fc / conv contractions.  The gaps after the mining's device-to-host copy are real:
that is the one point of a step where the host has to wait for the GPU.
"""
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sys

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if ("roi_align_fwd" in r["Kernel_Name"] or "roi_align_wino7_pair" in r["Kernel_Name"])]
# one full step: from the optimizer launch before the second-last ROIAlign forward to the one before the last
from _trace_util import step_marks
opt = step_marks(rows)
a = max(i for i in opt if i < marks[-2])
b = max(i for i in opt if i < marks[-1])
sel = rows[a + 1:b + 1]
t0 = int(sel[0]["Start_Timestamp"])
end = t0
gaps = []
prev = None
busy = 0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end:
        gaps.append((s - end, (end - t0) / 1e6, prev, r["Kernel_Name"][:60]))
    else:
        pass
    if e > end:
        busy += e - max(s, end)
        end = e
        prev = r["Kernel_Name"][:60]
total = end - t0
print("step %.3f ms, busy %.3f ms, idle %.3f ms in %d gaps, %d kernels" % (total / 1e6, busy / 1e6, (total - busy) / 1e6, len(gaps), len(sel)))
for g in sorted(gaps, reverse=True)[:top]:
    print("%8.1f us at %7.3f ms   after %-60s before %s" % (g[0] / 1e3, g[1], g[2], g[3]))
