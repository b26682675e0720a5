"""Candidate stage A/B on the GPU box: fp16 pairs (sim_topk_pair.hip) vs the f32-input MFMA stage (sim_topk_mfma.hip), interleaved in
one process.  python tools/perf_match_ab.py [n] [d] [nq,nq,...] [k]"""
import os, sys, time
import torch
sys.path.insert(0, ".")
from cslam_amd import nns_matching as nnm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nqs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1024, 4096, 100_000]
k = int(sys.argv[4]) if len(sys.argv) > 4 else 5
gen = torch.Generator(device="cuda").manual_seed(1234)
bank = torch.randn((n, d), generator=gen, device="cuda"); bank /= bank.norm(dim=1, keepdim=True)
nn = nnm.NearestNeighborsMatching(); nn.add_items_device(bank)
for nq in nqs:
    q = torch.randn((nq, d), generator=gen, device="cuda"); q /= q.norm(dim=1, keepdim=True)
    res = {}
    for rnd in range(3):
        for stage in ("pair", "f32"):
            os.environ["CSLAM_MFMA_STAGE1"] = stage
            t0 = time.perf_counter(); out = nn.search_device(q, k, mode=nnm.MODE_MFMA); torch.cuda.synchronize()
            t = time.perf_counter() - t0
            if rnd:
                res.setdefault(stage, []).append((t, nn.last_kernel_ms(), nn.last_stats()))
            if rnd == 2:
                res[stage + "_out"] = [x.clone() for x in out]
    fl = 2.0 * n * nq * d
    same = all(torch.equal(a, b) for a, b in zip(res["pair_out"], res["f32_out"]))
    rows_same = torch.equal(res["pair_out"][0], res["f32_out"][0]) and torch.equal(res["pair_out"][2], res["f32_out"][2])
    dmax = float((res["pair_out"][1] - res["f32_out"][1]).abs().max())
    for stage in ("pair", "f32"):
        t = min(x[0] for x in res[stage]); km = min(x[1] for x in res[stage])
        print(f"n={n} d={d} nq={nq} k={k} stage1={stage}: wall {t*1e3:.2f} ms ({nq/t:.0f} q/s)  stage-1 kernel {km:.2f} ms = "
              f"{fl/km/1e9:.1f} TFLOP/s f32-equivalent ({3*fl/km/1e9:.0f} fp16 TFLOP/s issued for pairs)  stats={res[stage][-1][2]}", flush=True)
    print(f"   results identical (rows, float64 scores, counts): {same}; rows and counts identical: {rows_same}; max |score difference| {dmax:.2e}", flush=True)
