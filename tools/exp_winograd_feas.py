#!/usr/bin/env python
"""Feasibility probe: fp32 strided-batched GEMM rates (the 16 Winograd F(2x2,3x3) GEMMs of VGG layers)
against MIOpen's direct fp32 convolution on the same layers, B = 128."""
import torch, time
import torch.nn.functional as F
dev = "cuda"
B = 128
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for name, hw, cin, cout in [("conv1_2", 224, 64, 64), ("conv2_2", 112, 128, 128), ("conv3_2", 56, 256, 256),
                            ("conv4_2", 28, 512, 512), ("conv5_2", 14, 512, 512)]:
    x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    tc = t(lambda: F.conv2d(x, w, padding=1))
    fl = 2.0 * B * hw * hw * cin * cout * 9
    T = B * (hw // 2) * (hw // 2)
    V = torch.randn(16, T, cin, device=dev)
    U = torch.randn(16, cin, cout, device=dev)
    tb = t(lambda: torch.bmm(V, U))
    Ut = U.transpose(1, 2).contiguous()
    tb2 = t(lambda: torch.bmm(V, Ut.transpose(1, 2)))
    flw = 2.0 * 16 * T * cin * cout
    traffic = (V.numel() * 2 + 16 * T * cout * 2) * 4 / 8e12
    print(f"{name}: direct {tc*1e3:.2f} ms ({fl/tc/1e12:.0f} TF) | winograd bmm NN {tb*1e3:.2f} ms ({flw/tb/1e12:.0f} TF) "
          f"NT {tb2*1e3:.2f} ms | transform traffic floor {traffic*1e3:.2f} ms -> est total {(min(tb,tb2)+traffic)*1e3:.2f} ms")
