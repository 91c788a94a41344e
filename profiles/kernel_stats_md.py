#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats CSV -> markdown table of the ALS kernels.
Usage: python3 profiles/kernel_stats_md.py <prof dir> <out.md> <build tag> <bench args...>"""
import csv, glob, os, sys

d, out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3], " ".join(sys.argv[4:])
f = max(glob.glob(f"{d}/*/*kernel_stats.csv"), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "anonymous namespace" in r["Name"]]
lines = [f"# build {tag}: ALS kernels of `bench.py {args}`", "",
         f"Command (GPU box): `rocprofv3 --kernel-trace --stats -d {d} -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary {args}` "
         "(warm-up + timed iterations; bench.py's data generation and torch's own kernels are left out of the table)", "",
         "| kernel | calls | total ms | avg ms | max ms |", "|---|---|---|---|---|"]
for r in rows:
    lines.append(f"| `{r['Name'][:96]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                 f"{float(r['AverageNs'])/1e6:.4f} | {float(r['MaxNs'])/1e6:.4f} |")
# per-dispatch durations of the dominant kernel in launch order (U-step and V-step launches alternate)
tr = glob.glob(f"{d}/*/*kernel_trace.csv")
if tr:
    calls = []
    for r in csv.DictReader(open(max(tr, key=os.path.getmtime))):
        if "k_row_tasks" in r["Kernel_Name"]:
            calls.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    calls.sort()
    if calls:
        lines += ["", "`k_row_tasks` dispatches in launch order (ms): " + " ".join(f"{t:.3f}" for _, t in calls)]
open(out, "w").write("\n".join(lines) + "\n")
print(out, len(rows), "kernels")
