"""BASELINE config 4 on ONE GPU (bank-major): 8 robots x 50k x 4096 banks, every robot's 50k keyframes
matched best-1 against each of the 7 other banks (lcsm.py:45-53) + top-5 against its own bank.
python tools/perf_c4.py [rows_per_robot]"""
import sys, time
import torch
sys.path.insert(0, ".")
from cslam_amd import nns_matching as nnm

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
R, D = 8, 4096
banks, descs = [], []
for r in range(R):
    g = torch.Generator(device="cuda").manual_seed(1234 + r)
    b = torch.randn((N, D), generator=g, device="cuda"); b /= b.norm(dim=1, keepdim=True)
    nn = nnm.NearestNeighborsMatching(); nn.add_items_device(b)
    banks.append(nn); descs.append(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
kms = 0.0
nmatch = 0
for r in range(R):                       # bank-major: bank r stays hot while all other robots' queries stream
    for o in range(R):
        k = 5 if o == r else 1
        rows, sims, cnt = banks[r].search_device(descs[o], k, mode=nnm.MODE_MFMA)
        kms += banks[r].last_kernel_ms()
        if o != r:
            nmatch += int((sims[:, 0] >= 0.07).sum())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
fl = 2.0 * R * R * N * N * D
print(f"C4 on 1 GPU: {R} robots x {N} keyframes, {R*R} bank passes: {dt:.2f}s = {R*N/dt:.0f} keyframes/s "
      f"(each matched against all {R} banks); MFMA kernels {kms/1e3:.2f}s = {fl/kms/1e9:.1f} TFLOP/s "
      f"({fl/kms/1e9/157.3*100:.1f}% of fp32 MFMA peak); inter-robot matches >= 0.07: {nmatch}")
