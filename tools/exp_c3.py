#!/usr/bin/env python
"""conv1_1 (3 -> 64, 224x224) at B = 256: hand-written kernel vs torch conv + fused bias/ReLU pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from cslam_amd.vpr.winograd import WinogradTrunk
torch.backends.cudnn.benchmark = True
seq = nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.ReLU()).cuda().eval()
x = torch.randn((256, 3, 224, 224), device="cuda")
wt = WinogradTrunk(seq)
print([s.kind for s in wt.steps])
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
xl = x.contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    print(f"hand-written c3: {t(lambda: wt(x)):.2f} ms   torch conv+relu (NHWC): {t(lambda: seq(xl)):.2f} ms   "
          f"write floor {256*224*224*64*4/8e12*1e3:.2f} ms")
    print("max diff", (wt(x) - seq(xl)).abs().max().item())
