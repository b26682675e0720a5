#!/usr/bin/env python
"""Do the HBM-bound early layers of the Winograd trunk run faster in sub-chunks small enough for V / M to stay in
the 256 MB Infinity Cache?  Times conv1_1..pool1 (and ..pool2) of VGG-16 per frame at several batch sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from cslam_amd.vpr.netvlad import NetVLAD
from cslam_amd.vpr.winograd import WinogradTrunk
nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096}, None)
mods = list(nv.encoder)
total = 256
x = torch.randn((total, 3, 224, 224), device="cuda")
for name, upto in (("conv1_1..pool1", 5), ("conv1_1..pool2", 10), ("whole trunk", len(mods))):
    sub = nn.Sequential(*mods[:upto])
    print(name, [type(m).__name__ for m in sub][-3:])
    for B in (2, 4, 8, 16, 32, 64, 256):
        tr = WinogradTrunk(sub, 64, 4)
        def run():
            for s in range(0, total, B):
                tr(x[s:s + B])
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"  B={B:4d}: {min(ts)*1e3:7.2f} ms per {total} frames")
