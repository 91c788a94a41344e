#!/bin/bash
# Phase ablation of als_row_solve (ALS_ABLATE bit0: no Gram, bit1: no Cholesky panels, bit2: no
# transposed solve).  Outputs are wrong by construction; only the phase timings matter.   ablate.sh [dir of a tree]
cd ${1:-.}
for a in 0 1 2 4 6 7; do
  ALS_ABLATE=$a python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $ABLATE_ARGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('ablate=$a', 'user %.2f item %.2f gs %.2f stats %.2f' % (p['row_solve_user'], p['row_solve_item'], p['gs_sweep'], p['residual_stats']))"
done
