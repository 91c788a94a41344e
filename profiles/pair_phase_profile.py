#!/usr/bin/env python3
"""Where the two-waves-per-row kernel (row_pair.hip) spends its cycles, per phase and per wave role.

Needs the profiling build:  bash profiles/ab_builds.sh prof "-DPAIR_PROFILE"
Run:  ALS_HIP_LIB=collaborative-filtering_amd/csrc/libals_hip_prof.so python3 profiles/pair_phase_profile.py [k]
Synthetic row sets at k (default 128): U-like (plain solve, rows of ~130 ratings), V-like (factor-only with Gram
by-product, rows of ~600 ratings); every set also timed with ALS_ROW_PAIR=0 semantics (scratch withheld).
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collaborative_filtering_amd import layout            # noqa: E402
from collaborative_filtering_amd.als import _side_to_dev, _tasks_to_dev   # noqa: E402
from collaborative_filtering_amd.backend import HipBackend  # noqa: E402

PH = ["top barrier", "gram", "xbuf barrier", "dump+rhs", "barrier", "panel", "panel barrier", "update",
      "update barrier", "factor out", "epilogue", "transposed solve"]


def run(k, nrows, lo, hi, factor, be, dev, label):
    rng = np.random.default_rng(1)
    ncols = 50000
    lens = rng.integers(lo, hi, size=nrows)
    indptr = np.zeros(nrows + 1, np.int64)
    indptr[1:] = np.cumsum(lens)
    idx = rng.integers(0, ncols, size=int(indptr[-1])).astype(np.int32)
    vals = rng.uniform(0.5, 5, size=idx.size).astype(np.float32)
    side = layout.SparseSide(nrows, ncols, indptr, idx, vals)
    t = layout.build_row_tasks(side.indptr)
    sd, td = _side_to_dev(side, dev), _tasks_to_dev(t, dev)
    ld = layout.padded_k(k)
    f32 = torch.float32
    F = torch.zeros(ncols + 1, ld, dtype=f32, device=dev)
    F[:ncols, :k] = torch.randn(ncols, k, device=dev) * 0.3
    z = lambda *s: torch.zeros(*s, dtype=f32, device=dev)   # noqa: E731
    X, b, status = z(nrows, ld), z(nrows), torch.zeros(1, dtype=torch.int32, device=dev)
    kw = dict(k=k, ld=ld, side=sd, F=F, zero_row=ncols, bias_self=z(nrows), bias_other=z(ncols),
              mu=torch.tensor([3.0], dtype=torch.float64, device=dev), lam=2.0, lam_row=None, lam_b=1.0, lam_b_row=None,
              rhs_extra=None, diag_extra=None, X_out=X, bias_out=b, gram_out=None, factor_out=None, rhs_out=None,
              colsum_out=None, sumr_out=None, status=status, tasks=td, workspace=None)
    if factor:
        kw.update(gram_out=z(nrows, ld, ld), factor_out=z(nrows, ld, ld), rhs_out=z(nrows, ld), colsum_out=z(nrows, ld),
                  sumr_out=z(nrows), sumr2_out=z(nrows))
    lib = be.lib
    have_prof = hasattr(lib, "als_pair_profile_read")
    out = (C.c_ulonglong * 32)()
    for pair in (True, False):
        be.row_pair = pair
        for _ in range(2):
            be.row_solve(**kw)
        torch.cuda.synchronize()
        if have_prof:
            lib.als_pair_profile_read(out)
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            be.row_solve(**kw)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{label}: k={k} rows={nrows} len {lo}..{hi} pair={pair}: {ms:.3f} ms/launch, status {int(status.item())}")
        if pair and have_prof:
            lib.als_pair_profile_read(out)
            a = np.array(list(out), dtype=np.float64).reshape(2, 16) / n
            ntask = t.tasks.shape[0] - t.ndual - t.nmid
            for w in (0, 1):
                tot = a[w].sum()
                print(f"   wave {'AB'[w]}: total {tot / ntask:9.0f} cycles/task; " +
                      ", ".join(f"{PH[i]} {a[w][i] / ntask:7.0f}" for i in range(len(PH))))


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda", 0)
    be = HipBackend(dev)
    run(k, 16384, 100, 160, False, be, dev, "U-like")
    run(k, 16384, 500, 700, True, be, dev, "V-like")
    run(k, 16384, 100, 160, True, be, dev, "V-like short")
    run(k, 8192, 900, 1100, True, be, dev, "V-like long")
    run(k, 8192, 900, 1100, False, be, dev, "U-like long")


if __name__ == "__main__":
    main()
