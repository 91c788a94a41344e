#!/usr/bin/env python3
"""Copy the round-2 evidence from gpurun_out/ into profiles/ (tracked):
  gpurun_out/<tag>_bench.json, prof_<tag>/ (rocprofv3 --kernel-trace --stats), <tag>_sq*/tcc* (PMC passes via
  collect_pmc.sh + pmc_to_json.py), other sizes -> profiles/<tag>_{bench.json, kernel_stats.md, pmc_k_row_tasks.json,
  other_sizes.json}.   Usage: python3 profiles/assemble_r02.py r02z"""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)


def last_json(fn):
    return json.loads(open(fn).read().strip().splitlines()[-1])


b = last_json(f"gpurun_out/{tag}_bench.json")
json.dump(b, open(f"profiles/{tag}_bench.json", "w"), indent=1)
other = {}
for fn in sorted(glob.glob(f"gpurun_out/{tag}_size_*.json")):
    name = os.path.basename(fn)[len(tag) + 6:-5]
    try:
        j = last_json(fn)
    except Exception:       # noqa: BLE001
        continue
    other[name] = {k: j[k] for k in ("value", "ms_per_step", "phase_ms_per_step")}
    other[name]["workload"] = j["config"]["workload"]
    other[name]["setup_s"] = j["config"]["setup_s"]
    other[name]["train_rmse_last"] = j["train_rmse"][-1]
if other:
    json.dump({"what": "bench.py --size <s> --steps 5 --warmup 2 --no-cpu-baseline on one MI355X (diagnostic sizes; "
                       "f64 = --solve-dtype float64, product = --graph product)", "runs": other},
              open(f"profiles/{tag}_other_sizes.json", "w"), indent=1)
stats = glob.glob(f"gpurun_out/prof_{tag}/*/*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(max(stats, key=os.path.getmtime))))
    out = [f"# round 2, build {tag}: per-kernel times of the headline run", "",
           f"Command (GPU box): `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_{tag} -- python3 bench.py "
           "--steps 5 --warmup 2 --no-cpu-baseline`", "",
           f"bench.py line of an un-profiled run of the same build: `{tag}_bench.json` (roofline.avg_launch_ms is the k_row_tasks "
           "average over the U-step and V-step launches; both launches are the same kernel, 7 iterations x 2 = 14 calls).", "",
           "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
    for r in rows[:18]:
        out.append(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                   f"{float(r['AverageNs']) / 1e6:.4f} | {r['Percentage']} |")
    out += ["", "ALS kernels: k_row_tasks<4,1> (U-step and V-step launch of every iteration; KB = 4, MODE = 1: bf16x3 Gram), "
            "k_sum_slots + k_row_long (split rows), k_gs_dataflow (one launch per iteration), k_sum_pairs_partial / "
            "k_sumsq4_partial / k_history_final / k_reduce_final (statistics).  Everything else is bench.py's synthetic-data "
            "generation, outside the timed region."]
    open(f"profiles/{tag}_kernel_stats.md", "w").write("\n".join(out) + "\n")
    for r in rows[:40]:
        if "k_row_tasks" in r["Name"] or "k_gs_dataflow" in r["Name"]:
            print(r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6)
if os.path.exists(f"gpurun_out/{tag}_pmc_k_row_tasks.json"):
    os.replace(f"gpurun_out/{tag}_pmc_k_row_tasks.json", f"profiles/{tag}_pmc_k_row_tasks.json")
r = b["roofline"]
print("cfg4", b["value"], b["ms_per_step"], b["phase_ms_per_step"], "frac", r["frac"], "frac_iter", r["frac_iter"],
      "avg_launch_ms", r["avg_launch_ms"], "cpu", (b.get("cpu_baseline") or {}).get("value"))
for k, v in other.items():
    print(k, v["value"], v["ms_per_step"], v["phase_ms_per_step"], v["setup_s"])
