#!/usr/bin/env python3
"""Copy one evidence refresh from gpurun_out/ into profiles/ under a round tag.
Usage: python3 profiles/assemble_round.py r01g   (expects gpurun_out/<tag>_bench.json, prof_<tag'>/, ...)"""
import csv, glob, json, os, sys

tag = sys.argv[1]                      # e.g. r01g
short = tag.replace("r0", "r")         # rocprof directory: prof_r1g
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
b = json.loads(open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1])
json.dump(b, open(f"profiles/{tag}_bench.json", "w"))
other = {}
for sname in ("cfg2", "cfg3", "cfg5-small", "k80", "k96", "k160"):
    fn = f"gpurun_out/{tag}_{sname}.json"
    if not os.path.exists(fn):
        continue
    j = json.loads(open(fn).read().strip().splitlines()[-1])
    other[sname] = {k: j[k] for k in ("value", "ms_per_step", "phase_ms_per_step", "train_rmse")}
    other[sname]["workload"] = j["config"]["workload"]
json.dump({"what": f"bench.py --size <s> --steps 5 --warmup 2 --no-cpu-baseline on one MI355X, build {tag} "
                   "(diagnostic sizes)", "runs": other}, open(f"profiles/{tag}_other_sizes.json", "w"), indent=1)
f = max(glob.glob(f"gpurun_out/prof_{short}/*/*kernel_stats.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
out = [f"# round 1, build {tag} (bf16x3 Gram, blocked MFMA Cholesky, fused statistics, dataflow sweep with publication buffer)", "",
       f"Command (GPU box): `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_{short} -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline`", "",
       f"bench.py line of an un-profiled run of the same build: `{tag}_bench.json` (roofline.avg_launch_ms is the k_row_tasks average over the U-step and V-step launches).", "",
       "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
for r in rows[:16]:
    out.append(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e6:.4f} | {r['Percentage']} |")
out += ["", "ALS kernels: k_row_tasks<4,1> (U-step and V-step launch of every iteration; KB=4, MODE=1: bf16x3 Gram), k_sum_slots + k_row_long (split rows), k_gs_dataflow (one launch per iteration), k_sum_pairs_partial / k_sumsq_partial / k_reduce_final (statistics).  Everything else is bench.py's synthetic-data generation, outside the timed region."]
open(f"profiles/{tag}_kernel_stats.md", "w").write("\n".join(out) + "\n")
os.replace(f"gpurun_out/{tag}_pmc_k_row_tasks.json", f"profiles/{tag}_pmc_k_row_tasks.json") if os.path.exists(f"gpurun_out/{tag}_pmc_k_row_tasks.json") else None
with open(f"profiles/{tag}_ablation.txt", "w") as fh:
    fh.write(f"# phase ablation of als_row_solve at cfg4 (build {tag}; profiles/ablate.sh; outputs wrong by construction, timings only)\n")
    fh.write(open(f"gpurun_out/{tag}_ablation.txt").read())
print("cfg4", b["value"], b["ms_per_step"], b["phase_ms_per_step"], "frac", b["roofline"]["frac"], "avg_launch_ms",
      b["roofline"]["avg_launch_ms"], b["roofline"]["traffic_source"], "cpu", b["cpu_baseline"]["value"], b["speedup_vs_cpu_baseline"])
for k, v in other.items():
    print(k, v["value"], v["ms_per_step"], v["phase_ms_per_step"])
for r in rows[:30]:
    if "k_row_tasks" in r["Name"] or "k_gs_dataflow" in r["Name"]:
        print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6)
