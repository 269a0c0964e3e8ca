#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED steps only, from a rocprofv3 --kernel-trace CSV of bench.py.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline
    python3 tools/trace_steps.py /tmp/kt/r_kernel_trace.csv 4 > profiles/r1/bench_kernel_stats.csv

A step is delimited by its (single) ROIAlign forward dispatch; everything before the first of the last
`steps` ROIAlign forwards (warm-up, MIOpen's one-off solver search on a box with a cold database, the
one-off mask-IoU maps) is dropped.  Columns mirror rocprofv3 --stats, plus per-step figures."""
import collections
import csv
import os
import sys


def product_name(r):
    """The pair GEMM serves nine products per step under three instantiations: one row per PRODUCT, told apart by the launch's
    workgroup count (batched Winograd products are multiples of the 121 positions: forward 4 N-tiles per M-tile, data gradient 8;
    launches of exactly 256 workgroups of the M-contiguous-A instantiations are the chunked late weight gradients; fc1's forward
    product at <= 1024 proposals also has 256 workgroups - 4 x 16 tiles x 4 K-splits - under <0, 0>), so that the roofline's dominant launch - the
    Winograd-domain forward GEMM - has its own average duration in this file."""
    name = r["Kernel_Name"]
    if "gemm_pair_kernel" not in name:
        return name
    if "Grid_Size" in r:
        wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
    else:       # kernel-trace CSV: per-dimension columns (work-items)
        g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        w = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
        wgs = g // max(w, 1)
    if wgs % 121 == 0:
        per = wgs // 121
        tag = "Winograd forward" if "<0, 0" in name else "Winograd data gradient" if "<0, 1" in name else "Winograd weight gradient"
        return "%s [%s GEMM: 121 x %d tiles]" % (name, tag, per)
    if wgs == 256 and "<1, " in name:          # (weight gradients contract over the rows of both operands: M-contiguous A)
        return "%s [late weight-gradient chunk: 256 workgroups]" % name
    return "%s [fc product: %d workgroups]" % (name, wgs)


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if ("roi_align_fwd" in r["Kernel_Name"] or "roi_align_wino7_pair" in r["Kernel_Name"])]
    assert len(marks) >= steps, "fewer ROIAlign forwards than steps in the trace"
    # the backbone forward precedes ROIAlign inside a step: start at the end of the previous step's last kernel,
    # i.e. right after the optimizer's last dispatch before the first timed ROIAlign -> use the previous ROIAlign
    # forward's step end; simplest robust cut: begin at the first dispatch after the (steps+1)-th last mark's step.
    first = marks[-steps]
    # walk back to the start of that step: the backbone's first kernel follows the previous step's optimizer
    # (multi_tensor_apply) - find the last optimizer dispatch before `first`
    # (a step BEGINS with its first optimizer dispatch - the previous step's update; with optim.SGD.overlap_update the big weights'
    # launch follows on the side stream and runs under this step's backbone forward: it is counted with the step it overlaps)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _trace_util import step_marks
    opt_marks = step_marks(rows)
    start = max([m for m in opt_marks if m < first], default=0)
    # ... and end with the optimizer launch that closes the last step (what follows is bench.py's own end-of-run work: its check
    # that every parameter is finite - 2 launches per tensor - and the extra loops)
    end = max([m for m in opt_marks if m > start], default=len(rows))           # the first optimizer dispatch behind the last step
    sel = rows[start:end]
    stat = collections.OrderedDict()
    for r in sel:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        s = stat.setdefault(product_name(r), [0, 0, 1 << 62, 0])
        s[0] += 1
        s[1] += d
        s[2] = min(s[2], d)
        s[3] = max(s[3], d)
    total = sum(s[1] for s in stat.values())
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "CallsPerStep", "MsPerStep"])
    for name, s in sorted(stat.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name, s[0], s[1], "%.1f" % (s[1] / s[0]), "%.3f" % (100.0 * s[1] / total), s[2], s[3],
                    "%.2f" % (s[0] / steps), "%.4f" % (s[1] / steps / 1e6)])
    w.writerow(["TOTAL (%d timed steps)" % steps, sum(s[0] for s in stat.values()), total, "", "100", "", "", "",
                "%.4f" % (total / steps / 1e6)])


if __name__ == "__main__":
    main()
