#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs written by collect_pmc.sh: per kernel name and
counter, the mean value per dispatch (ALS kernels only)."""
import csv, glob, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "pmc1"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"gpurun_out/{tag}_*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "k_row" not in name and "k_gs" not in name and "k_residual" not in name:
            continue
        short = name.split("::")[-1].split("(")[0]
        acc[short][(r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("==", k)
    for (c, g), v in sorted(acc[k].items()):
        print(f"   {c:32s} grid={g:>10s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
