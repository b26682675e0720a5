"""Does the match leg (fp32-MFMA bound) hide under the extract leg (mostly HBM bound) when they run on two streams?
Two Python threads (ctypes and torch release the GIL in their calls): one extracts 1024 frames per step, the other searches
1024 queries against the 100k bank per step; against the same work back to back."""
import sys, threading, time
import torch
sys.path.insert(0, ".")
from cslam_amd import nns_matching as nnm
from cslam_amd.vpr.netvlad import NetVLAD

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
N, B, CH, D, ROWS, K = 6, 1024, 512, 4096, 100000, 5
gen = torch.Generator(device=dev).manual_seed(1234)
bank = torch.randn((ROWS, D), generator=gen, device=dev)
bank /= bank.norm(dim=1, keepdim=True)
nn = nnm.NearestNeighborsMatching(device=0)
nn.add_items_device(bank)
ex = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": D,
              "frontend.random_seed": 0}, None)
frames = torch.randint(0, 256, (B, 480, 640, 3), device=dev, dtype=torch.uint8)
q = torch.randn((B, D), device=dev)
q /= q.norm(dim=1, keepdim=True)


def extract():
    return torch.cat([ex.compute_embeddings_device(frames[s:s + CH], None) for s in range(0, B, CH)])


def search():
    return nn.search_device(q, K, mode=nnm.MODE_MFMA)


extract(); search(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    extract()
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N):
    search()
torch.cuda.synchronize(); tm = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N):
    extract(); search()
torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / N
s2 = torch.cuda.Stream(device=dev)


def worker():
    torch.cuda.set_device(dev)
    with torch.cuda.stream(s2):
        for _ in range(N):
            search()
    s2.synchronize()


th = threading.Thread(target=worker)
t0 = time.perf_counter()
th.start()
for _ in range(N):
    extract()
torch.cuda.synchronize()
th.join()
tc = (time.perf_counter() - t0) / N
print(f"per step of {B}: extract {te*1e3:.2f} ms, match {tm*1e3:.2f} ms, back to back {ts*1e3:.2f} ms, on two streams {tc*1e3:.2f} ms "
      f"-> {B/ts:.0f} vs {B/tc:.0f} keyframes/s", flush=True)
