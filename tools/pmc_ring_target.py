"""Launch target for rocprofv3 --pmc passes over the candidate-stage variants (tools/gpu_ring_pmc.sh): 100k x 4096 bank, searches of
NQ queries with CSLAM_MFMA_RING = each of VARIANTS (measurement build), REPS launches each."""
import os
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd import nns_matching as nnm

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1, 0, 1]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
os.environ["CSLAM_MFMA_STAGE1"] = "h1"
gen = torch.Generator(device="cuda").manual_seed(1234)
bank = torch.randn((100_000, 4096), generator=gen, device="cuda")
bank /= bank.norm(dim=1, keepdim=True)
nn = nnm.NearestNeighborsMatching()
nn.add_items_device(bank)
q = torch.randn((nq, 4096), generator=gen, device="cuda")
q /= q.norm(dim=1, keepdim=True)
for v in variants:
    os.environ["CSLAM_MFMA_RING"] = str(v)
    for _ in range(reps):
        nn.search_device(q, 5, mode=nnm.MODE_MFMA)
        torch.cuda.synchronize()
    print(v, nn.last_stats(), nn.last_kernel_ms(), flush=True)
