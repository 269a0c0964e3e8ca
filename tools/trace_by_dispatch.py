#!/usr/bin/env python3
"""Mean kernel duration per (kernel, grid) RUN of a rocprofv3 --kernel-trace CSV, in dispatch order: consecutive dispatches of the
same kernel and grid are one run (a micro-benchmark's timing loop); prints runs with >= MIN dispatches."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Dispatch_Id"]))
minn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
runs = []
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if "Grid_Size" in r:
        wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
    else:
        gsz = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        wsz = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
        wgs = gsz // max(wsz, 1)
    key = (name, wgs)
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if runs and runs[-1][0] == key:
        runs[-1][1].append(d)
    else:
        runs.append((key, [d]))
# a timing loop of TWO kernels (product + reduce) alternates: merge alternating pairs
merged = []
i = 0
while i < len(runs):
    if len(runs[i][1]) == 1 and i + 3 < len(runs) and runs[i + 2][0] == runs[i][0] and runs[i + 3][0] == runs[i + 1][0]:
        a, b = runs[i][0], runs[i + 1][0]
        da, db = [], []
        while i + 1 < len(runs) and runs[i][0] == a and runs[i + 1][0] == b and len(runs[i][1]) == 1:
            da += runs[i][1]; db += runs[i + 1][1]; i += 2
        merged.append((a, da)); merged.append((b, db))
    else:
        merged.append(runs[i]); i += 1
for key, ds in merged:
    if len(ds) >= minn:
        ds = ds[len(ds) // 4:]
        print("%-60s wgs=%6d n=%3d mean=%7.2f us min=%7.2f" % (key[0][:60], key[1], len(ds), sum(ds) / len(ds) / 1e3, min(ds) / 1e3))
