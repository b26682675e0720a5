"""BASELINE config 2 on one GPU: CosPlace ResNet-18 512-D extract of 10k synthetic 640x480 keyframes +
causal intra-robot NNS (top-5 of every keyframe among the EARLIER keyframes) over the growing bank.
python tools/perf_c2.py [frames] [chunk] [winograd|winograd2|direct]"""
import sys, time
import torch
sys.path.insert(0, ".")
from cslam_amd import nns_matching as nnm
from cslam_amd.vpr.cosplace import CosPlace

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 250
MODE = sys.argv[3] if len(sys.argv) > 3 else "winograd"
torch.backends.cudnn.benchmark = True
cp = CosPlace({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376,
               "frontend.cosplace.descriptor_dim": 512, "frontend.cosplace.backbone": "resnet18",
               "frontend.backbone_conv": MODE}, None)
g = torch.Generator(device="cuda").manual_seed(7)
frames = torch.randint(0, 256, (CH, 480, 640, 3), generator=g, device="cuda", dtype=torch.uint8)
cp.compute_embeddings_device(frames); torch.cuda.synchronize()          # warm-up (MIOpen find)
nn = nnm.NearestNeighborsMatching()
t0 = time.perf_counter(); te = 0.0; tm = 0.0
done = 0
while done < N:
    m = min(CH, N - done)
    a = time.perf_counter()
    d = cp.compute_embeddings_device(frames[:m]); torch.cuda.synchronize()
    b = time.perf_counter()
    nn.add_items_device(d)                                               # descriptors never leave HBM
    lim = torch.arange(done, done + m, device="cuda", dtype=torch.int64) # keyframe i sees rows < i
    rows, sims, cnt = nn.search_device(d, 5, row_limit=lim, mode=nnm.MODE_AUTO); torch.cuda.synchronize()
    c = time.perf_counter()
    te += b - a; tm += c - b; done += m
dt = time.perf_counter() - t0
# the same with the search in two halves: finish() of chunk i behind the enqueue of chunk i + 1's extraction (bench.py's c2 leg)
nn2 = nnm.NearestNeighborsMatching()
torch.cuda.synchronize()
t1 = time.perf_counter(); pend = None; done2 = 0
while done2 < N:
    m = min(CH, N - done2)
    d = cp.compute_embeddings_device(frames[:m])
    if pend is not None:
        pend.finish()
    nn2.add_items_device(d)
    lim = torch.arange(done2, done2 + m, device="cuda", dtype=torch.int64)
    pend = nn2.search_device_async(d, 5, row_limit=lim, mode=nnm.MODE_AUTO)
    done2 += m
rows2, sims2, cnt2 = pend.finish(); torch.cuda.synchronize()
dp = time.perf_counter() - t1
assert torch.equal(rows2, rows) and torch.equal(cnt2, cnt)
print(f"C2 [{MODE}] pipelined: {N / dp:.0f} keyframes/s")
# two extraction lanes: 2 x CH frames per iteration, the passes alternating over two streams
big = torch.cat([frames, frames])
dl = cp.compute_embeddings_batch_device(big, CH, 2); torch.cuda.synchronize()
assert torch.equal(dl[:CH], cp.compute_embeddings_device(frames))
nn3 = nnm.NearestNeighborsMatching()
torch.cuda.synchronize()
t2 = time.perf_counter(); pend = None; done3 = 0
while done3 < N:
    m = min(2 * CH, N - done3)
    d = cp.compute_embeddings_batch_device(big[:m], CH, 2)
    if pend is not None:
        pend.finish()
    nn3.add_items_device(d)
    lim = torch.arange(done3, done3 + m, device="cuda", dtype=torch.int64)
    pend = nn3.search_device_async(d, 5, row_limit=lim, mode=nnm.MODE_AUTO)
    done3 += m
pend.finish(); torch.cuda.synchronize()
dl2 = time.perf_counter() - t2
print(f"C2 [{MODE}] pipelined, two extraction lanes: {N / dl2:.0f} keyframes/s")
print(f"C2 [{MODE}]: {N} keyframes, chunk {CH}: extract+match {N/dt:.0f} keyframes/s (extract {N/te:.0f}/s, causal match {N/tm:.0f}/s), "
      f"bank rows {nn.n}, last chunk cnt min {int(cnt.min())}")
