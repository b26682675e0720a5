"""Per-kernel timing of the descriptor heads at batch B (GPU box): python tools/perf_heads.py [B]"""
import sys, time
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr import heads

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(0)

def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

frames = torch.randint(0, 256, (B, 480, 640, 3), generator=g, device="cuda", dtype=torch.uint8)
t = bench(lambda: heads.preprocess(frames, 376))
byt = B * (376 * 376 * 3 + 2 * 376 * 224 * 3 + 224 * 224 * 3 * 4)
print(f"preprocess   B={B}: {t*1e6/B:8.2f} us/frame  {byt/t/1e9:8.1f} GB/s (algorithmic: crop read + u8 intermediate rw + f32 write)")
f = torch.randn((B, 512, 14, 14), generator=g, device="cuda")
w = torch.randn((64, 512), generator=g, device="cuda"); c = torch.rand((64, 512), generator=g, device="cuda")
t = bench(lambda: heads.vlad_aggregate(f, w, None, c))
byt = B * (512 * 196 * 4 + 64 * 512 * 4)
print(f"vlad         B={B}: {t*1e6/B:8.2f} us/frame  {byt/t/1e9:8.1f} GB/s  {B*25.7e6/t/1e12:6.2f} TFLOP/s")
v = torch.randn((B, 32768), generator=g, device="cuda")
comp = torch.randn((4096, 32768), generator=g, device="cuda") / 181.0
t = bench(lambda: heads.pca_project(v, comp, None, None), 5)
print(f"pca 32768->4096 B={B}: {t*1e6/B:8.2f} us/frame  {2.0*B*32768*4096/t/1e12:6.1f} TFLOP/s  weights {4096*32768*4/t/1e9:7.1f} GB/s")
f7 = torch.randn((B, 512, 7, 7), generator=g, device="cuda").abs()
W = torch.randn((512, 512), generator=g, device="cuda") / 22.0; b = torch.zeros(512, device="cuda")
t = bench(lambda: heads.gem_fc_head(f7, 3.0, 1e-6, W, b))
byt = B * (512 * 49 * 4 + 512 * 4)
print(f"gem+fc       B={B}: {t*1e6/B:8.2f} us/frame  {byt/t/1e9:8.1f} GB/s")
x = torch.randn((B, 4096), generator=g, device="cuda")
t = bench(lambda: heads.l2_normalize_(x))
print(f"l2norm 4096  B={B}: {t*1e6/B:8.2f} us/row    {B*4096*8/t/1e9:8.1f} GB/s")
