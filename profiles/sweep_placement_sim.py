#!/usr/bin/env python3
"""Event model of the dataflow Gauss-Seidel sweep (k_gs_dataflow) on the benchmark graph (cfg 4: 100 000 items, top-50
of 512 candidates, ~45 dependencies per item, 266 levels): what would a CHAIN-AWARE placement of items on waves buy?

Today items are dealt round-robin in (level, id) order, so every hop of every dependency chain is a hand-off through
memory (`remote`, ~3.5 us: sc1 store -> sc1 poll across XCDs).  If an item ran in the workgroup that solved its LATEST
dependency, that one value could travel through LDS (`local`, ~0.3 us; the publication buffer stays the fallback).
finish(i) = max(wave_free + pre, max_d(finish(d) + latency(d -> i))) + solve; `pre` = what a wave needs between two
of its items before it can solve (item id -> S_ptr -> S_idx chunk -> first poll of the published rows -> FMAs).

    python profiles/sweep_placement_sim.py > profiles/r03_sweep_placement_sim.txt        (CPU only, ~2 min)
"""
import heapq
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                                  # noqa: E402

S = bench.gen_graph(torch.device("cpu"), 100_000, seed=2004)
ptr, idx = S[0].numpy(), S[1].numpy()
n = len(ptr) - 1
deps = [idx[ptr[i]:ptr[i + 1]][idx[ptr[i]:ptr[i + 1]] < i] for i in range(n)]
level = np.zeros(n, np.int64)
for i in range(n):
    if deps[i].size:
        level[i] = level[deps[i]].max() + 1
order = np.lexsort((np.arange(n), level))
W, G = 2048, 4                        # resident waves, waves per workgroup
uniq = np.mean([(level[d] == level[d].max()).sum() == 1 for d in deps if d.size])
print(f"{n} items, {level.max() + 1} levels, {np.mean([d.size for d in deps]):.1f} dependencies per item; "
      f"{uniq:.2f} of the items have ONE dependency on the level just below their own")


def place(policy):
    wave_of = np.full(n, -1)
    wlastlev = np.full(W, -1)
    wlast = np.full(W, -1)
    cnt = np.zeros(W, np.int64)
    heap = [(0, w) for w in range(W)]
    heapq.heapify(heap)
    rr = 0
    for i in order:
        d, w = deps[i], -1
        if policy in ("wave", "workgroup") and d.size:
            lv = level[d]
            for j in d[lv == lv.max()][::-1]:
                if policy == "wave":
                    if wlast[wave_of[j]] == j:
                        w = wave_of[j]
                        break
                else:
                    ws = np.arange(wave_of[j] // G * G, wave_of[j] // G * G + G)
                    c = ws[np.argmin(wlastlev[ws])]
                    if wlastlev[c] < level[i]:
                        w = c
                        break
        if w < 0:
            if policy == "round-robin":
                w, rr = rr % W, rr + 1
            else:
                while True:
                    c, w = heapq.heappop(heap)
                    if c == cnt[w]:
                        break
        wave_of[i], wlastlev[w], wlast[w] = w, level[i], i
        cnt[w] += 1
        heapq.heappush(heap, (cnt[w], w))
    return wave_of, cnt


def sim(wave_of, same, solve=1.2, remote=3.5, local=0.3, pre=3.0):
    fin, wfree = np.zeros(n), np.zeros(W)
    for i in order:
        d, w = deps[i], wave_of[i]
        ready = (fin[d] + np.where(same(wave_of[d], w), local, remote)).max() if d.size else 0.0
        fin[i] = max(wfree[w] + pre, ready) + solve
        wfree[w] = fin[i]
    return fin.max()


for pol, same in (("round-robin", lambda a, b: a == b), ("wave", lambda a, b: a == b),
                  ("workgroup", lambda a, b: a // G == b // G)):
    wo, cnt = place(pol)
    print(f"{pol:12s} items per wave {cnt.min()}..{cnt.max()}: " +
          "  ".join(f"pre {pre:.1f} us -> {sim(wo, same, pre=pre):7.1f} us" for pre in (0.6, 2.0, 3.0, 5.0)))
