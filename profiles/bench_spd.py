"""Timing of als_spd_solve_f64 against the library Cholesky it replaced (torch.linalg.cholesky_ex +
cholesky_solve = rocSOLVER potrf/potrs), W-step sized systems.  Usage: python profiles/bench_spd.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collaborative_filtering_amd.backend import HipBackend

dev = torch.device("cuda", 0)
be = HipBackend(dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
for N in (64, 256, 640, 1216, 2432, 4864):
    rng = np.random.default_rng(N)
    B = rng.normal(size=(N, N // 2 + 1))
    A = torch.from_numpy(B @ B.T + N * np.eye(N)).to(dev)
    b = torch.from_numpy(rng.normal(size=N)).to(dev)

    def ours():
        return be.spd_solve(A, b, 0.5, status)

    def lib():
        L, info = torch.linalg.cholesky_ex(A + 0.5 * torch.eye(N, dtype=torch.float64, device=dev))
        return torch.cholesky_solve(b.reshape(N, 1), L).reshape(N)

    out = {}
    for name, fn in (("hip", ours), ("library", lib)):
        for _ in range(3):
            x = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            x = fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = (e0.elapsed_time(e1) / 10, x)
    err = float((out["hip"][1] - out["library"][1]).norm() / out["library"][1].norm())
    print(f"N={N:5d}  als_spd_solve_f64 {out['hip'][0]:8.3f} ms   library potrf+potrs {out['library'][0]:8.3f} ms   "
          f"rel diff {err:.2e}  status {int(status.item())}", flush=True)
