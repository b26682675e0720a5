"""Candidate stage A/B on the GPU box: one fp16 product on the hi halves (h1, the default), exact fp16 pairs (pair: three products) and
the f32-input MFMA stage (f32), interleaved in one process.  python tools/perf_match_ab.py [n] [d] [nq,nq,...] [k] [stages]"""
import os, sys, time
import torch
sys.path.insert(0, ".")
from cslam_amd import nns_matching as nnm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nqs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1024, 4096, 100_000]
k = int(sys.argv[4]) if len(sys.argv) > 4 else 5
stages = sys.argv[5].split(",") if len(sys.argv) > 5 else ["h1", "pair", "f32"]
PROD = {"h1": 1, "pair": 3, "f32": 0}
gen = torch.Generator(device="cuda").manual_seed(1234)
bank = torch.randn((n, d), generator=gen, device="cuda"); bank /= bank.norm(dim=1, keepdim=True)
nn = nnm.NearestNeighborsMatching(); nn.add_items_device(bank)
for nq in nqs:
    q = torch.randn((nq, d), generator=gen, device="cuda"); q /= q.norm(dim=1, keepdim=True)
    res = {}
    for rnd in range(3):
        for stage in stages:
            os.environ["CSLAM_MFMA_STAGE1"] = stage
            t0 = time.perf_counter(); out = nn.search_device(q, k, mode=nnm.MODE_MFMA); torch.cuda.synchronize()
            t = time.perf_counter() - t0
            if rnd:
                res.setdefault(stage, []).append((t, nn.last_kernel_ms(), nn.last_stats()))
            if rnd == 2:
                res[stage + "_out"] = [x.clone() for x in out]
    fl = 2.0 * n * nq * d
    ref = stages[-1]
    for stage in stages:
        t = min(x[0] for x in res[stage]); km = min(x[1] for x in res[stage])
        same = all(torch.equal(a, b) for a, b in zip(res[stage + "_out"], res[ref + "_out"]))
        dmax = float((res[stage + "_out"][1] - res[ref + "_out"][1]).abs().max())
        issued = f"{PROD[stage]*fl/km/1e9:.0f} fp16 TFLOP/s issued" if PROD[stage] else "f32-input MFMA"
        print(f"n={n} d={d} nq={nq} k={k} stage1={stage}: wall {t*1e3:.2f} ms ({nq/t:.0f} q/s)  stage-1 kernel {km:.3f} ms = "
              f"{fl/km/1e9:.1f} TFLOP/s f32-equivalent ({issued})  stats={res[stage][-1][2]}  "
              f"identical to {ref} (rows, float64 scores, counts): {same}, max |score difference| {dmax:.2e}", flush=True)
