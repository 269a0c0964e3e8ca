#!/usr/bin/env python3
"""Phases of the last complete training step in a rocprofv3 --kernel-trace CSV of bench.py (main stream, wall ms):
optimizer | backbone forward | ROIAlign forward | MaskFuse forward + heads | mining + losses | heads + MaskFuse backward |
ROIAlign backward | backbone backward.

    python3 tools/trace_phases_step.py /tmp/kt/r_kernel_trace.csv"""
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


def first(pat, after=0):
    for i, r in enumerate(step):
        if i >= after and pat in r["Kernel_Name"]:
            return i
    return None


def last(pat):
    idx = [i for i, r in enumerate(step) if pat in r["Kernel_Name"]]
    return idx[-1] if idx else None


marks = [("optimizer", 0, 0)]
i_tab = first("roi_tables")
i_fwd = last("roi_align_wino7_pair") if last("roi_align_wino7_pair") is not None else last("roi_align_fwd")
i_loss = first("losses_kernel")
i_seed = first("step_mine_kernel")
i_rb0 = first("roi_align_bwd")
i_rb1 = last("roi_partial_reduce") or last("roi_align_bwd")
pts = [("optimizer launch", ms(step[0]["Start_Timestamp"]), ms(step[0]["End_Timestamp"])),
       ("backbone forward", ms(step[0]["End_Timestamp"]), ms(step[i_tab]["Start_Timestamp"])),
       ("ROIAlign forward", ms(step[i_tab]["Start_Timestamp"]), ms(step[i_fwd]["End_Timestamp"])),
       ("MaskFuse forward + heads", ms(step[i_fwd]["End_Timestamp"]), ms(step[i_seed]["Start_Timestamp"])),
       ("mining + losses", ms(step[i_seed]["Start_Timestamp"]), ms(step[i_loss]["End_Timestamp"])),
       ("heads + MaskFuse backward", ms(step[i_loss]["End_Timestamp"]), ms(step[i_rb0]["Start_Timestamp"])),
       ("ROIAlign backward", ms(step[i_rb0]["Start_Timestamp"]), ms(step[i_rb1]["End_Timestamp"])),
       ("backbone backward (+ host gap before the optimizer)", ms(step[i_rb1]["End_Timestamp"]), ms(step[-1]["Start_Timestamp"]))]
for name, s, e in pts:
    print("%-55s %7.3f ms   (%.3f .. %.3f)" % (name, e - s, s, e))
print("%-55s %7.3f ms" % ("step", ms(step[-1]["Start_Timestamp"])))
