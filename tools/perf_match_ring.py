"""A/B of the candidate stage on 256 x 256 tiles (GPU box, measurement build):
    CSLAM_HIP_LIB=cslam_amd/libcslam_hip_abl.so python tools/perf_match_ring.py [nq,nq,...] [variants] [dbgs] [reps]
variants: -1 = one workgroup per work item (sim_topk_pair_kernel), 0..3 = persistent schedule (sim_topk_ring.hip; bit 0 = tile-start
rendezvous, bit 1 = static priority for waves 4..7).  dbgs: 0 = product, 1 = no global loads (timing only), 2 = every request an L2
hit (timing only).  All variants of a (nq, dbg = 0) row must return identical results; rounds are interleaved (guide rule 24)."""
import os
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd import nns_matching as nnm

nqs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [100_000, 1024]
variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1, 0, 1, 2, 3]
dbgs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n, d, k = 100_000, 4096, 5
os.environ.setdefault("CSLAM_MFMA_STAGE1", "h1")      # no back-off to the f32 stage after the (wrong by design) DBG results
gen = torch.Generator(device="cuda").manual_seed(1234)
bank = torch.randn((n, d), generator=gen, device="cuda")
bank /= bank.norm(dim=1, keepdim=True)
nn = nnm.NearestNeighborsMatching()
nn.add_items_device(bank)
for nq in nqs:
    q = torch.randn((nq, d), generator=gen, device="cuda")
    q /= q.norm(dim=1, keepdim=True)
    fl = 2.0 * n * nq * d
    for dbg in dbgs:
        os.environ["CSLAM_MFMA_DBG"] = str(dbg)
        ms = {v: [] for v in variants}
        ref = None
        for r in range(reps + 1):
            for v in variants:
                os.environ["CSLAM_MFMA_RING"] = str(v)
                out = nn.search_device(q, k, mode=nnm.MODE_MFMA)
                torch.cuda.synchronize()
                if r > 0:
                    ms[v].append(nn.last_kernel_ms())
                if dbg == 0 and r == 0:
                    got = tuple(t.clone() for t in out)
                    if ref is None:
                        ref = got
                    else:
                        same = all(torch.equal(a, b) or torch.equal(a.nan_to_num(7.0), b.nan_to_num(7.0)) for a, b in zip(ref, got))
                        print(f"  nq={nq} variant {v}: results identical to variant {variants[0]}: {same}  stats={nn.last_stats()}")
        for v in variants:
            best, med = min(ms[v]), sorted(ms[v])[len(ms[v]) // 2]
            print(f"nq={nq} dbg={dbg} variant={v:2d}: kernel min {best:.3f} ms = {fl / best / 1e9:.1f} TF  median {med:.3f} ms = "
                  f"{fl / med / 1e9:.1f} TF  stats={nn.last_stats()}", flush=True)
