#!/bin/bash
# bench.py on the production library and on each A/B build (profiles/ab_builds.sh), same box, back to back.
#   profiles/ab_bench.sh out_prefix tag1 tag2 ...     (extra bench.py flags via $BENCH_FLAGS)
out=$1; shift
cs=collaborative-filtering_amd/csrc
for rep in 1 2; do
  for tag in base "$@"; do
    lib=$cs/libals_hip.so; [ "$tag" != base ] && lib=$cs/libals_hip_$tag.so
    ALS_HIP_LIB=$PWD/$lib python bench.py --steps 8 --warmup 2 --no-cpu-baseline $BENCH_FLAGS > ${out}_${tag}_$rep.json 2> ${out}_${tag}_$rep.err || echo "FAILED $tag"
    python - <<PY
import json
d=json.load(open("${out}_${tag}_$rep.json")); p=d["phase_ms_per_step"]
print("$tag rep $rep: %.3f ms/iter  U %.3f  V %.3f  sweep %.3f  rmse %.9f" % (d["ms_per_step"], p["row_solve_user"], p["row_solve_item"], p.get("gs_sweep",0), d["train_rmse"][-1]))
PY
  done
done
