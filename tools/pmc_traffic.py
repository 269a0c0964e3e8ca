#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into per-kernel HBM traffic per launch.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_FETCH_SIZE -o r -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_WRITE_SIZE -o r -- python3 bench.py ...
    python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE <conv_algo> <engine> <workload tag> \
            [--calibrate /tmp/cal_FETCH_SIZE /tmp/cal_WRITE_SIZE]  > profiles/r3/pmc_traffic_<config>.json

Counters are in KiB per dispatch.  MI355X_MICROARCH.md (HBM) warns that on gfx950 FETCH_SIZE can report half of the
bytes of a wide streaming read and prescribes a calibration on a known byte count with the kernel's own access width.
--calibrate takes the two passes of tools/pmc_calibrate.py (1 GiB copied at 16 B per lane; 256 MiB of 4-byte strided
accesses) and stores bytes_counted / bytes_moved for reads and writes; the per-kernel figures below are divided by the
16-byte factors (all product kernels listed here read and write 16 B per lane)."""
import collections
import csv
import json
import sys

# product kernels -> the key bench.py looks up
KEYS = (("gemm_pair_kernel<0, 0, false>", "wino_gemm_fwd"), ("gemm_pair_kernel<0, 0>", "wino_gemm_fwd"),
        ("roi_align_wino7_pair", "cim_roi_align_wino7_pair_fwd"), ("wino7_dx_maskfold", "wino7_dx_maskfold"),
        ("wino7_flatten_bwd_dy_pair", "wino7_flatten_bwd_dy_pair"), ("step_mine_kernel", "step_mine"),
        ("roi_align_fwd_rowsum", "cim_roi_align_maskcat_fwd"), ("roi_align_fwd_agg", "cim_roi_align_maskcat_fwd"),
        ("roi_align_bwd_region_kernel<false", "cim_roi_align_bwd"), ("roi_align_bwd_region", "cim_roi_align_maskcat_bwd"),
        ("roi_partial_reduce", "roi_partial_reduce"), ("mask_iou_pair", "mask_iou_pair"), ("mask_pack", "mask_pack"),
        ("sgd_multi", "sgd_multi"), ("gemm_small_kernel", "backbone_conv1x1"))


def per_kernel(d, counter, names=None):
    rows = csv.DictReader(open(d + "/r_counter_collection.csv"))
    vals = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            try:
                wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            except (KeyError, ValueError):
                wgs = 0
            vals[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), wgs))
    return vals


def calibrate(fdir, wdir):
    f, w = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    out = {}
    for tag, needle, nbytes in (("copy16", "vectorized_elementwise_kernel", float(1 << 30)), ("copy4", "elementwise_kernel", None)):
        pick = lambda k: needle in k and "FillFunctor" not in k and ("vectorized" in needle or "vectorized" not in k)
        fk = [v[1] for k, vs in f.items() if pick(k) for v in vs]
        wk = [v[1] for k, vs in w.items() if pick(k) for v in vs]
        if not fk or not wk:
            continue
        big_f, big_w = max(fk), max(wk)
        if nbytes is None:      # strided 4-byte copy: every 64-byte sector of 256 MiB x 2 is touched; bytes USED are 128 MiB
            out[tag] = dict(fetch_kib=big_f, write_kib=big_w, note="2^25 4-byte elements at stride 8 B: 128 MiB used, 256 MiB of lines")
        else:
            out[tag] = dict(fetch_kib=big_f, write_kib=big_w, fetch_factor=big_f * 1024 / nbytes, write_factor=big_w * 1024 / nbytes)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    fetch, write = per_kernel(args[0], "FETCH_SIZE"), per_kernel(args[1], "WRITE_SIZE")
    out = {"_conv_algo": args[2] if len(args) > 2 else "winograd7", "_engine": args[3] if len(args) > 3 else "f16x2",
           "_workload": args[4] if len(args) > 4 else "mix8"}
    ff = wf = 1.0
    if "--calibrate" in sys.argv:
        i = sys.argv.index("--calibrate")
        cal = calibrate(sys.argv[i + 1], sys.argv[i + 2])
        out["_calibration"] = cal
        if "copy16" in cal:
            ff, wf = cal["copy16"]["fetch_factor"], cal["copy16"]["write_factor"]
    out["_applied_factors"] = dict(fetch=ff, write=wf)
    for needle, key in KEYS:
        fl = [v for k, vs in fetch.items() if needle in k for v in sorted(vs)]
        wl = [v for k, vs in write.items() if needle in k for v in sorted(vs)]
        if not fl or key in out:
            continue
        if key == "wino_gemm_fwd":
            # this instantiation also runs fc products and late weight-gradient chunks: the Winograd-domain forward GEMM of the
            # MaskFuse conv is the launch whose workgroup count is a multiple of the 121 positions (121 x 16 / 121 x 20 tiles)
            if any(v[2] for v in fl):
                fl = [v for v in fl if v[2] and v[2] % 121 == 0]
                wl = [v for v in wl if v[2] and v[2] % 121 == 0]
            else:
                fl, wl = fl[0::3], wl[0::3]
        f, w = [v[1] for v in fl], [v[1] for v in wl]
        # (the first launches of a run belong to the warm-up: same kernels, same sizes)
        out[key] = dict(kernel=needle, dispatches=len(f), fetch_kib_mean=sum(f) / len(f), write_kib_mean=(sum(w) / len(w)) if w else None,
                        hbm_bytes_mean=(sum(f) / len(f) / ff + ((sum(w) / len(w) / wf) if w else 0)) * 1024)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
