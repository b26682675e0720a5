#!/usr/bin/env python
"""Winograd trunk vs the direct (MIOpen) trunk: agreement and throughput on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd.vpr.netvlad import NetVLAD
from cslam_amd.vpr.winograd import WinogradTrunk

torch.backends.cudnn.benchmark = True
nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096}, None)
def t(fn, n=3):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
g = torch.Generator(device="cuda").manual_seed(0)
for minc, tile in ((128, 2), (128, 4), (64, 4)):
    wt = WinogradTrunk(nv.encoder, min_in_channels=minc, tile=tile)
    x = torch.randn((8, 3, 224, 224), generator=g, device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        a = nv.encoder(x); b = wt(x)
        r = nv.encoder.double()(x.double()); nv.encoder.float()
    err = (a - b).abs().max().item() / a.abs().max().item()
    e64w = (b.double() - r).abs().max().item() / r.abs().max().item()
    e64d = (a.double() - r).abs().max().item() / r.abs().max().item()
    print(f"tile {tile}: vs float64 trunk: winograd {e64w:.2e}, direct fp32 {e64d:.2e}")
    print(f"min_in_channels {minc}: wino steps {sum(s.kind == 'wino' for s in wt.steps)}, max |diff| / max |ref| = {err:.2e}")
    for B in (1, 128, 256):
        x = torch.randn((B, 3, 224, 224), generator=g, device="cuda").contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            td = t(lambda: nv.encoder(x)); tw = t(lambda: wt(x))
        print(f"   B={B}: direct {td*1e3:.2f} ms ({B/td:.0f} frames/s)   winograd {tw*1e3:.2f} ms ({B/tw:.0f} frames/s)")
