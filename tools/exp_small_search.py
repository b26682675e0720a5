#!/usr/bin/env python
"""Cost of one chunk-sized search (250 queries) against robot-sized banks, by mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import nns_matching as nnm
for n in (2500, 12500, 125000):   # scan mode with >= 16 queries = the float64 tile kernel
    bank = torch.randn((n, 4096), device="cuda"); bank /= bank.norm(dim=1, keepdim=True)
    m = nnm.NearestNeighborsMatching(); m.add_items_device(bank)
    for dt in (torch.float32, torch.float64):
        q = (bank[:250] + 0.01 * torch.randn((250, 4096), device="cuda")).to(dt)
        for mode, name in ((nnm.MODE_MFMA, "mfma"), (nnm.MODE_SCAN, "scan"), (nnm.MODE_AUTO, "auto")):
            for k in (1, 10):
                for _ in range(3): m.search_device(q, k, mode=mode)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): m.search_device(q, k, mode=mode)
                torch.cuda.synchronize(); dt_ms = (time.perf_counter() - t0) / 10 * 1e3
                print(f"n={n:6d} {str(dt)[6:]:8s} {name:5s} k={k:2d}: {dt_ms:7.3f} ms  (kernel {m.last_kernel_ms():.3f} ms, stats {m.last_stats()})")
