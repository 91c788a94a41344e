#!/bin/bash
# Interleaved A/B of bench.py builds on ONE box (cdna guide rule 24): profiles/ab_bench.sh <outdir> <reps> name=ENVSPEC ...
#   ENVSPEC: "lib:<path to libals_hip_*.so>" | "tree:<dir holding an older copy of the repo>" | "base"
# Prints per-variant phase times; JSON lines go to <outdir>/ab_<name>_<rep>.json.
OUT=$1; REPS=$2; shift 2
mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for spec in "$@"; do
    name=${spec%%=*}; what=${spec#*=}
    case $what in
      lib:*)  ( export ALS_HIP_LIB=$PWD/${what#lib:}; python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $AB_ARGS > $OUT/ab_${name}_$rep.json 2> $OUT/ab_${name}_$rep.err ) ;;
      tree:*) ( cd ${what#tree:} && python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $AB_TREE_ARGS > $OLDPWD/$OUT/ab_${name}_$rep.json 2> $OLDPWD/$OUT/ab_${name}_$rep.err ) ;;
      *)      python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $AB_ARGS > $OUT/ab_${name}_$rep.json 2> $OUT/ab_${name}_$rep.err ;;
    esac
    python3 - $OUT/ab_${name}_$rep.json $name $rep <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = d["phase_ms_per_step"]
    print("%-8s rep %s: iter %.3f ms  user %.3f item %.3f gs %.3f  rmse %.9f" % (sys.argv[2], sys.argv[3], d["ms_per_step"],
          p.get("row_solve_user", 0), p.get("row_solve_item", 0), p.get("gs_sweep", 0), d["train_rmse"][-1]), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
  done
done
