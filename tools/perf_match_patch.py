"""XCD patch shape of the candidate stage's work-item order (CSLAM_MFMA_PATCH = query tiles x bank segments per XCD), 100k x 100k x 4096,
interleaved rounds.  python tools/perf_match_patch.py"""
import os, sys, time, statistics
import torch
sys.path.insert(0, ".")
from cslam_amd import nns_matching as nnm
n = nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000; d = 4096
nq = int(sys.argv[2]) if len(sys.argv) > 2 else nq
gen = torch.Generator(device="cuda").manual_seed(1234)
bank = torch.randn((n, d), generator=gen, device="cuda"); bank /= bank.norm(dim=1, keepdim=True)
nn = nnm.NearestNeighborsMatching(); nn.add_items_device(bank)
q = torch.randn((nq, d), generator=gen, device="cuda"); q /= q.norm(dim=1, keepdim=True)
shapes = ["default", "6,6", "4,8", "8,4", "2,16", "16,2", "1,32", "32,1", "3,11", "11,3"]
res = {s: [] for s in shapes}
for rnd in range(3):
    for s in shapes:
        if s == "default": os.environ.pop("CSLAM_MFMA_PATCH", None)
        else: os.environ["CSLAM_MFMA_PATCH"] = s
        nn.search_device(q, 5, mode=nnm.MODE_MFMA); torch.cuda.synchronize()
        if rnd: res[s].append(nn.last_kernel_ms())
fl = 2.0 * n * nq * d
for s in shapes:
    km = statistics.median(res[s])
    print(f"n={n} nq={nq} stage1={os.environ.get('CSLAM_MFMA_STAGE1', 'h1')} patch {s:8s}: stage-1 kernel {km:7.3f} ms = {fl/km/1e9:6.1f} TFLOP/s (2 D flop per pair)", flush=True)
