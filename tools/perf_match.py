"""Quick match-throughput probe (GPU box): python tools/perf_match.py [n] [d] [nq] [k]"""
import sys, time
import torch
sys.path.insert(0, ".")
from cslam_amd import nns_matching as nnm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nqs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4096, 100_000]
k = int(sys.argv[4]) if len(sys.argv) > 4 else 5
gen = torch.Generator(device="cuda").manual_seed(1234)
bank = torch.randn((n, d), generator=gen, device="cuda"); bank /= bank.norm(dim=1, keepdim=True)
nn = nnm.NearestNeighborsMatching(); nn.add_items_device(bank)
for nq in nqs:
    q = torch.randn((nq, d), generator=gen, device="cuda"); q /= q.norm(dim=1, keepdim=True)
    out = nn.search_device(q, k, mode=nnm.MODE_MFMA); torch.cuda.synchronize()
    ts, ks = [], []
    for _ in range(3):
        t0 = time.perf_counter(); nn.search_device(q, k, mode=nnm.MODE_MFMA, out=out); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0); ks.append(nn.last_kernel_ms())
    t, km = min(ts), min(ks)
    fl = 2.0 * n * nq * d
    print(f"n={n} d={d} nq={nq} k={k}: wall {t*1e3:.1f} ms ({nq/t:.0f} q/s)  mfma kernel {km:.1f} ms "
          f"= {fl/km/1e9:.1f} TFLOP/s ({fl/km/1e9/157.3*100:.1f}% of 157.3)  stats={nn.last_stats()}")
for nq in (1, 4, 8):
    q = torch.randn((nq, d), generator=gen, device="cuda"); q /= q.norm(dim=1, keepdim=True)
    out = nn.search_device(q, k, mode=nnm.MODE_SCAN); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); nn.search_device(q, k, mode=nnm.MODE_SCAN, out=out); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"scan nq={nq}: {t*1e6:.0f} us  -> {n*d*4/t/1e12*((nq+3)//4):.2f} TB/s bank stream")
