"""Launch target for the rocprofv3 --pmc passes of the matcher's candidate stage (tools/gpu_pmc_match.sh): a 100k x 4096 bank and, in
order, REPS searches of 100 000 queries (BASELINE config 3's batch) and REPS of 1024 (the bench's in-step launch), top-5, the default
candidate stage (one fp16 product); CSLAM_MFMA_STAGE1=pair / f32 profile the other stages instead."""
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd import nns_matching as nnm

REPS = 3
gen = torch.Generator(device="cuda").manual_seed(1234)
bank = torch.randn((100_000, 4096), generator=gen, device="cuda")
bank /= bank.norm(dim=1, keepdim=True)
nn = nnm.NearestNeighborsMatching()
nn.add_items_device(bank)
for nq in (100_000, 1024):
    q = torch.randn((nq, 4096), generator=gen, device="cuda")
    q /= q.norm(dim=1, keepdim=True)
    for _ in range(REPS):
        nn.search_device(q, 5, mode=nnm.MODE_MFMA)
    torch.cuda.synchronize()
    print(nq, nn.last_stats(), nn.last_kernel_ms())
