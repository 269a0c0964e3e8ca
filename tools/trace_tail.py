#!/usr/bin/env python3
"""End of a training step in a rocprofv3 --kernel-trace CSV of bench.py: for the last complete step, when each HIP queue
(stream) runs its last kernel before the optimizer launch, and what runs after the ROIAlign backward - i.e. whether the
backbone backward (main stream) or the weight-gradient GEMMs (side stream) end the step.

    python3 tools/trace_tail.py /tmp/kt/r_kernel_trace.csv"""
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
from _trace_util import step_marks
opt = step_marks(rows)
a, b = opt[-2], opt[-1]
step = rows[a:b + 1]
t0 = int(step[0]["Start_Timestamp"])
ms = lambda t: (int(t) - t0) / 1e6
print("step: %.3f ms from optimizer launch to optimizer launch" % ms(step[-1]["Start_Timestamp"]))
roi = [r for r in step if "roi_align_bwd" in r["Kernel_Name"]]
t_roi = int(roi[-1]["End_Timestamp"]) if roi else t0
print("ROIAlign backward ends at %.3f ms" % ms(t_roi))
queues = {}
for r in step[1:-1]:
    queues.setdefault(r["Queue_Id"], []).append(r)
for q, rs in queues.items():
    last = max(rs, key=lambda r: int(r["End_Timestamp"]))
    busy_after = sum(min(int(r["End_Timestamp"]), 1 << 62) - max(int(r["Start_Timestamp"]), t_roi) for r in rs if int(r["End_Timestamp"]) > t_roi)
    print("queue %s: %4d kernels, last ends at %.3f ms (%s), busy after the ROIAlign backward %.3f ms" % (
        q, len(rs), ms(last["End_Timestamp"]), last["Kernel_Name"][:50], busy_after / 1e6))
print("kernels > 0.3 ms ending after the ROIAlign backward:")
for r in step[1:-1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 0.3 and int(r["End_Timestamp"]) > t_roi:
        print("   %.3f .. %.3f ms  queue %s  %s" % (ms(r["Start_Timestamp"]), ms(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"][:70]))
