#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc CSVs of collect_pmc.sh into profiles/<round>_pmc_k_row_tasks.json:
per-launch means of every counter for the two k_row_tasks launches of an iteration (told apart by
grid size: the larger grid is the U-step), the corrected HBM bytes and their mean (bench.py's
roofline.traffic).  Usage: python3 profiles/pmc_to_json.py <tag> <out.json>"""
import collections, csv, glob, json, sys

tag, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"gpurun_out/{tag}_*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_row_tasks" not in r["Kernel_Name"]:
            continue
        acc[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
grids = sorted(acc, reverse=True)
assert len(grids) == 2, grids
res = {}
for name, g in zip(("user_step_launch", "item_step_launch"), grids):
    d = {c: sum(v) / len(v) for c, v in sorted(acc[g].items())}
    d["hbm_bytes_corrected"] = 2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024
    res[name] = d
res["traffic_bytes_per_launch_mean"] = 0.5 * (res["user_step_launch"]["hbm_bytes_corrected"]
                                              + res["item_step_launch"]["hbm_bytes_corrected"])
res["note"] = ("rocprofv3 --pmc passes of profiles/collect_pmc.sh (separate passes, --kernel-trace only) on "
               "`bench.py --steps 2 --warmup 1 --no-cpu-baseline`; means over the dispatches of each grid size; "
               "FETCH_SIZE/WRITE_SIZE are in KB; hbm_bytes_corrected = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 "
               "(MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950)")
json.dump(res, open(out, "w"), indent=1)
print(out, res["traffic_bytes_per_launch_mean"])
