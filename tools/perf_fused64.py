"""conv1_2 of VGG-16 (64 -> 64 channels, 224 x 224, + ReLU + MaxPool) at 256 frames: fused F(2x2) kernel vs the
F(4x4) transform / rocBLAS / transform pipeline; and the whole trunk either way."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from cslam_amd.vpr.netvlad import NetVLAD
from cslam_amd.vpr.winograd import WinogradTrunk
nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096}, None)
mods = list(nv.encoder)
def best(fn, n=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)
x1 = torch.randn((256, 64, 224, 224), device="cuda").contiguous(memory_format=torch.channels_last)
sub = nn.Sequential(*mods[2:5])
print([type(m).__name__ for m in sub])
for f in (False, True):
    tr = WinogradTrunk(sub, 64, 4, fused64=f)
    print(f"conv1_2+relu+pool fused64={f}: {best(lambda: tr(x1))*1e3:.3f} ms per 256 frames")
del x1
x = torch.randn((256, 3, 224, 224), device="cuda")
for f in (False, True):
    tr = WinogradTrunk(nv.encoder, 64, 4, fused64=f)
    print(f"whole trunk fused64={f}: {best(lambda: tr(x))*1e3:.3f} ms per 256 frames")
del x
x1 = torch.randn((256, 64, 224, 224), device="cuda").contiguous(memory_format=torch.channels_last)
for waves in ("4", "0"):
    os.environ["CSLAM_WF_WAVES"] = waves
    tr = WinogradTrunk(sub, 64, 4, fused64=True)
    print(f"conv1_2+relu+pool fused waves={waves}: {best(lambda: tr(x1))*1e3:.3f} ms per 256 frames")
del x1
x2 = torch.randn((256, 64, 112, 112), device="cuda").contiguous(memory_format=torch.channels_last)
sub2 = nn.Sequential(*mods[5:7])
print([type(m).__name__ for m in sub2], mods[5])
for couts, waves in (("64", "4"), ("64,128", "4"), ("64,128", "8"), ("64,128", "0")):
    os.environ["CSLAM_WINO_FUSED_COUTS"] = couts; os.environ["CSLAM_WF_WAVES"] = waves
    tr = WinogradTrunk(sub2, 64, 4, fused64=True)
    print(f"conv2_1+relu fused couts={couts} waves={waves}: {best(lambda: tr(x2))*1e3:.3f} ms per 256 frames")
del x2
x = torch.randn((256, 3, 224, 224), device="cuda")
for couts, waves in (("64", "4"), ("64,128", "4"), ("64", "0"), ("64,128", "0")):
    os.environ["CSLAM_WINO_FUSED_COUTS"] = couts; os.environ["CSLAM_WF_WAVES"] = waves
    tr = WinogradTrunk(nv.encoder, 64, 4, fused64=True)
    print(f"whole trunk fused couts={couts} waves={waves}: {best(lambda: tr(x))*1e3:.3f} ms per 256 frames")
for ft in ("2", "4"):
    os.environ["CSLAM_WINO_FUSED_TILE"] = ft; os.environ["CSLAM_WINO_FUSED_COUTS"] = "64,128"; os.environ["CSLAM_WF_WAVES"] = "0"
    tr = WinogradTrunk(nv.encoder, 64, 4, fused64=True)
    print(f"whole trunk fused tile F({ft}x{ft}): {best(lambda: tr(x))*1e3:.3f} ms per 256 frames")
del x
x1 = torch.randn((256, 64, 224, 224), device="cuda").contiguous(memory_format=torch.channels_last)
for ft in ("2", "4"):
    os.environ["CSLAM_WINO_FUSED_TILE"] = ft
    tr = WinogradTrunk(sub, 64, 4, fused64=True)
    print(f"conv1_2+relu+pool fused tile F({ft}x{ft}): {best(lambda: tr(x1))*1e3:.3f} ms per 256 frames")
del x1
x2 = torch.randn((256, 64, 112, 112), device="cuda").contiguous(memory_format=torch.channels_last)
for ft in ("2", "4"):
    os.environ["CSLAM_WINO_FUSED_TILE"] = ft
    tr = WinogradTrunk(sub2, 64, 4, fused64=True)
    print(f"conv2_1+relu fused tile F({ft}x{ft}): {best(lambda: tr(x2))*1e3:.3f} ms per 256 frames")
