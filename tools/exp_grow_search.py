#!/usr/bin/env python
"""Incremental pattern of the deployment: add a chunk of 250 rows, search 250 queries, repeat (bank grows to 12.5k)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import nns_matching as nnm
torch.manual_seed(0)
rows = torch.randn((12500, 4096), device="cuda"); rows /= rows.norm(dim=1, keepdim=True)
for label, qdt in (("f32", torch.float32), ("f64", torch.float64)):
    m = nnm.NearestNeighborsMatching()
    t_add = t_search = t_cpu = 0.0
    per = []
    for s in range(0, 12500, 250):
        q = rows[s:s + 250].to(qdt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.add_items_device(rows[s:s + 250])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out = m.search_device(q, 1)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        r = [o.cpu() for o in out]
        t3 = time.perf_counter()
        t_add += t1 - t0; t_search += t2 - t1; t_cpu += t3 - t2
        per.append((t2 - t1) * 1e3)
    print(f"{label}: 50 x (add 250, search 250): add {t_add/50*1e3:.3f} ms, search {t_search/50*1e3:.3f} ms, readback {t_cpu/50*1e3:.3f} ms; "
          f"search ms by bank size: first {per[0]:.2f}, n=2.5k {per[9]:.2f}, n=6k {per[23]:.2f}, n=12.5k {per[49]:.2f}")
    for _ in range(3): m.search_device(q, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.search_device(q, 1)
    torch.cuda.synchronize(); print(f"   static bank n=12.5k: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per search")
