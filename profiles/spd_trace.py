import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from collaborative_filtering_amd.backend import HipBackend
dev = torch.device("cuda", 0); be = HipBackend(dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
for N in (256, 1216):
    rng = np.random.default_rng(N); B = rng.normal(size=(N, N // 2 + 1))
    A = torch.from_numpy(B @ B.T + N * np.eye(N)).to(dev); b = torch.from_numpy(rng.normal(size=N)).to(dev)
    for _ in range(3): be.spd_solve(A, b, 0.5, status)
    torch.cuda.synchronize()
    T = (N + 63) // 64; NP = 64 * T
    off = 256 + 8 * ((T + 1) * 64 * NP + T * 4096)
    tr = be._spd_ws[off: off + 8 * 4 * T].view(torch.int64).cpu().numpy().reshape(T, 4)
    t0 = tr[0, 0]
    print("N", N, "per panel [panel, barrier1, update, barrier2(until next panel)] us:")
    for j in range(T):
        nxt = tr[j + 1, 0] if j + 1 < T else tr[j, 3]
        print(j, [(tr[j, 1] - tr[j, 0]) / 100, (tr[j, 2] - tr[j, 1]) / 100, (tr[j, 3] - tr[j, 2]) / 100, (nxt - tr[j, 3]) / 100])
    print("total factor us", (tr[-1, 3] - t0) / 100)
