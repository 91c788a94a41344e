#!/usr/bin/env python3
"""Per-item phase times of the dataflow Gauss-Seidel sweep from a profiling build of the library
(gs_sweep.hip with -DALS_GS_STAMPS, selected with ALS_HIP_LIB): s_memtime totals over all waves and items.
    ALS_HIP_LIB=.../libals_hip_stamps.so python profiles/sweep_phase_stamps.py [bench args]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from collaborative_filtering_amd import _hip                   # noqa: E402

out = bench.main(sys.argv[1:] or ["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-secondary"])
lib = _hip.load()
buf = (C.c_ulonglong * 8)()
assert lib.als_debug_gs_stamps(buf) == 0
n = max(buf[0], 1)
names = ["requests: descriptor, S_ptr, factor column, rhs (until all have landed)", "non-dependency gather (pass 1)",
         "dependency batches: polls + waits (pass 2)", "substitutions", "publication + bias / statistics epilogue"]
print(f"sweep {out['phase_ms_per_step']['gs_sweep']:.3f} ms per iteration; {buf[0]} item visits in the stamped launches")
for q, nm in enumerate(names, start=1):
    print(f"  {nm:75s} {buf[q] / n * 0.01:8.2f} us per item")
print(f"  {'sum':75s} {sum(buf[1:6]) / n * 0.01:8.2f} us per item per wave")
