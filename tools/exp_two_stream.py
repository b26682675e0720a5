#!/usr/bin/env python
"""Does running two half-batches of the Winograd trunk on two HIP streams overlap the HBM-bound transforms
of one with the MFMA-bound GEMMs of the other?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd.vpr.netvlad import NetVLAD
from cslam_amd.vpr.winograd import WinogradTrunk
nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096}, None)
B = 256
x = torch.randn((B, 3, 224, 224), device="cuda").contiguous(memory_format=torch.channels_last)
t1 = WinogradTrunk(nv.encoder, 64, 4)
def bench(fn, n=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
print(f"one stream, B={B}: {B / bench(lambda: t1(x)):.0f} frames/s")
for nsplit in (2, 4):
    trunks = [WinogradTrunk(nv.encoder, 64, 4) for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    parts = list(x.chunk(nsplit))
    def run():
        cur = torch.cuda.current_stream()
        for s in streams: s.wait_stream(cur)
        for tr, s, p in zip(trunks, streams, parts):
            with torch.cuda.stream(s):
                tr(p)
        for s in streams: cur.wait_stream(s)
    print(f"{nsplit} streams x B={B // nsplit}: {B / bench(run):.0f} frames/s")
