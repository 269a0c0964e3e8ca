#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc pass (SQ / GRBM counters) with the derived figures the roofline argument needs.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES \\
              SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc -o r -- python3 tools/bench_gemm_pair.py ...
    python3 tools/pmc_summary.py /tmp/pmc NEEDLE [NEEDLE ...]  > profiles/r4/pmc_<what>.json

For every kernel whose name contains a NEEDLE: dispatches, mean duration (kernel trace of the same pass), mean counter values per
dispatch, and
    mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (gui x 256 CUs x 4 SIMDs)                 (the gfx94x MfmaUtil formula)
    effective_clock  = gui / duration                                                      (MI355X_MICROARCH.md, DVFS)
  with gui = GRBM_GUI_ACTIVE / 8: rocprofv3 reports the counter SUMMED over the chip's 8 XCCs (a 2.4 GHz streaming kernel reads
  19.4 "GHz" otherwise; checked on the ROIAlign kernels, which are not power-limited).  SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles
  per v_mfma_f32_32x32x16_f16, summed over all SIMDs: mfma_busy_frac x effective_clock / 2.4 GHz reproduces the time-derived
  fraction of the 2516.6 TFLOP/s peak (0.64 x 1.82 / 2.4 = 0.485 for the Winograd forward GEMM).
    valu_per_mfma    = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA
    wave-cycle split = SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY (issue-stalled), SQ_WAIT_ANY (parked) over SQ_WAVE_CYCLES.
The profiler serialises kernels and runs the chip at a slightly lower clock than an un-profiled run: durations here are for the
ratios only."""
import collections
import csv
import glob
import json
import sys


def main():
    d, needles = sys.argv[1], sys.argv[2:]
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(cc)):
        name = r["Kernel_Name"]
        hit = [n for n in needles if n in name]
        if not hit:
            continue
        wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        key = "%s [%d workgroups]" % (short, wgs)
        agg[key][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
        meta[key] = dict(vgpr=r.get("VGPR_Count"), agpr=r.get("Accum_VGPR_Count"), lds=r.get("LDS_Block_Size"))
    out = {}
    for key, ctr in agg.items():
        ids = sorted({i for vs in ctr.values() for i, _ in vs}, key=int)
        ids = ids[len(ids) // 3:] if len(ids) >= 6 else ids               # (drop the first third: warm-up launches)
        mean = {c: sum(v for i, v in vs if i in ids) / max(1, sum(1 for i, _ in vs if i in ids)) for c, vs in ctr.items()}
        e = dict(dispatches=len(ids), counters=mean, **meta[key])
        ds = [dur[i] for i in ids if i in dur]
        if ds:
            e["duration_us"] = sum(ds) / len(ds) / 1e3
        g = mean.get("GRBM_GUI_ACTIVE")
        g = g / 8.0 if g else g                       # (summed over the 8 XCCs)
        if g and "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
            e["mfma_busy_frac"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 256 * 4)
        if g and ds:
            e["effective_clock_ghz"] = g / (sum(ds) / len(ds))
        if mean.get("SQ_INSTS_MFMA"):
            e["valu_per_mfma"] = (mean.get("SQ_INSTS_VALU", 0.0) - mean["SQ_INSTS_MFMA"]) / mean["SQ_INSTS_MFMA"]
        w = mean.get("SQ_WAVE_CYCLES")
        if w:
            for c, label in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue_stalled"), ("SQ_WAIT_ANY", "parked"),
                             ("SQ_ACTIVE_INST_VALU", "valu_active")):
                if c in mean:
                    e["wave_cycles_" + label] = mean[c] / w
        if g and "SQ_BUSY_CYCLES" in mean:
            e["sq_busy_cycles_over_gui"] = mean["SQ_BUSY_CYCLES"] / g
        out[key] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
