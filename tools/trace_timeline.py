#!/usr/bin/env python3
"""Dispatch-by-dispatch timeline of the last complete training step of a rocprofv3 --kernel-trace CSV of bench.py:
queue, start (ms from the optimizer launch that opens the step), duration (us), kernel.  --from PAT / --to PAT cut the window
to the first dispatch matching PAT .. the last matching PAT; --min-us hides short dispatches (they are summed per gap).

    python3 tools/trace_timeline.py /tmp/kt/r_kernel_trace.csv --from roi_align_fwd --to roi_align_bwd --min-us 20"""
import argparse
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import re

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--from", dest="frm", default=None)
ap.add_argument("--to", default=None)
ap.add_argument("--min-us", type=float, default=0.0)
args = ap.parse_args()
rows = list(csv.DictReader(open(args.trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
from _trace_util import step_marks
opt = step_marks(rows)
step = rows[opt[-2]:opt[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
lo, hi = 0, len(step) - 1
if args.frm:
    lo = next(i for i, r in enumerate(step) if args.frm in r["Kernel_Name"])
if args.to:
    hi = max(i for i, r in enumerate(step) if args.to in r["Kernel_Name"])
queues = {}
hidden_n, hidden_us = 0, 0.0


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]


for r in step[lo:hi + 1]:
    q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    us = (e - s) / 1e3
    if us < args.min_us:
        hidden_n += 1
        hidden_us += us
        continue
    if hidden_n:
        print("        ... %d short dispatches, %.0f us" % (hidden_n, hidden_us))
        hidden_n, hidden_us = 0, 0.0
    print("q%d %9.3f ms %9.1f us  %s" % (q, (s - t0) / 1e6, us, short(r["Kernel_Name"])))
print("step: %.3f ms" % ((int(step[-1]["Start_Timestamp"]) - t0) / 1e6))
