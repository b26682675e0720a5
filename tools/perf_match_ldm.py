"""Pair candidate stage: placement of the next stage's LDS-DMA requests (CSLAM_PAIR_LDM: 1 = all inside the first K step, 0 = half behind
each K step), interleaved.  python tools/perf_match_ldm.py"""
import os, sys, statistics
import torch
sys.path.insert(0, ".")
from cslam_amd import nns_matching as nnm
n, d = 100_000, 4096
gen = torch.Generator(device="cuda").manual_seed(1234)
bank = torch.randn((n, d), generator=gen, device="cuda"); bank /= bank.norm(dim=1, keepdim=True)
nn = nnm.NearestNeighborsMatching(); nn.add_items_device(bank)
for nq in (1024, 4096, 100_000):
    q = torch.randn((nq, d), generator=gen, device="cuda"); q /= q.norm(dim=1, keepdim=True)
    res = {"1": [], "0": []}; outs = {}
    for rnd in range(4):
        for ldm in ("1", "0"):
            os.environ["CSLAM_PAIR_LDM"] = ldm
            out = nn.search_device(q, 5, mode=nnm.MODE_MFMA); torch.cuda.synchronize()
            if rnd: res[ldm].append(nn.last_kernel_ms())
            outs[ldm] = [x.clone() for x in out]
    same = all(torch.equal(a, b) for a, b in zip(outs["1"], outs["0"]))
    fl = 2.0 * n * nq * d
    for ldm in ("1", "0"):
        km = statistics.median(res[ldm])
        print(f"nq={nq} LDM={ldm}: stage-1 kernel {km:.3f} ms = {fl/km/1e9:.1f} TFLOP/s fp32-equivalent ({3*fl/km/1e9:.0f} fp16)  stats={nn.last_stats()}", flush=True)
    print(f"   identical results: {same}", flush=True)
