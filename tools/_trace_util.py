"""Shared by the trace_* tools: where an optimizer step begins in a rocprofv3 kernel trace of bench.py."""
OPT = ("sgd_multi_kernel", "multi_tensor_apply", "FusedSgd")


def step_marks(rows, names=OPT, gap_ns=3_000_000):
    """Indices (rows sorted by start time) of the FIRST optimizer dispatch of every step: with cim_amd.optim.SGD.overlap_update a
    step has two sgd_multi_kernel launches - the small parameters on the step's stream, the big weights on the side stream right
    behind it; an optimizer dispatch within `gap_ns` of the previous one belongs to the same step."""
    marks, last = [], None
    for i, r in enumerate(rows):
        if any(t in r["Kernel_Name"] for t in names):
            s = int(r["Start_Timestamp"])
            if last is None or s - last > gap_ns:
                marks.append(i)
            last = s
    return marks
