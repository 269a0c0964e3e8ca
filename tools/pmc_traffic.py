#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into per-kernel HBM traffic.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_FETCH_SIZE -o r -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_WRITE_SIZE -o r -- python3 bench.py ...
    python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE [conv_algo] > profiles/r1/pmc_traffic.json

Counters are in KiB per dispatch.  MI355X_MICROARCH.md (HBM) warns that on gfx950 FETCH_SIZE can
report half of the bytes of a wide streaming read and asks for a calibration on a known byte
count in the kernel's own access pattern.  Calibration here: roi_align_bwd_tile_kernel reads
grad_out exactly once (N*49*2C*4 B = 383 MiB at cfg2) and reports FETCH_SIZE = 384 MiB;
roi_align_fwd / wino_input / wino_dy report WRITE_SIZE equal to their output size to the MiB.
So for these 16 B/lane, 64 B-segment access patterns no correction applies:
bytes = (FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import json
import sys

OURS = ("gemm_f32_kernel", "gemm_bf16x3_kernel", "gemm_f16x2_kernel", "amax_rowcol", "roi_align", "mask_iou", "mask_pack", "mask_area", "splitk", "asy_flag", "seed_select",
        "contain_argmax", "arbitrate", "assign_kernel", "wino")


def per_kernel(d, counter):
    rows = csv.DictReader(open(d + "/r_counter_collection.csv"))
    vals = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter and any(s in r["Kernel_Name"] for s in OURS):
            vals[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return vals


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    # which conv algorithm the profiled run used (bench.py only attaches GEMM traffic when it matches)
    out = {"_conv_algo": sys.argv[3] if len(sys.argv) > 3 else "winograd4"}
    for k in fetch:
        f = [v for _, v in sorted(fetch[k])]
        w = [v for _, v in sorted(write.get(k, []))]
        name = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()
        out[name] = dict(dispatches=len(f), fetch_kib_mean=sum(f) / len(f), write_kib_mean=(sum(w) / len(w)) if w else None,
                         fetch_kib_per_dispatch=f[:12], write_kib_per_dispatch=w[:12],
                         hbm_bytes_mean=(sum(f) / len(f) + ((sum(w) / len(w)) if w else 0)) * 1024)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
