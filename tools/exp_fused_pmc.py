"""conv1_2-shaped launches of the fused Winograd kernel in its forms (CSLAM_WF_WAVES = 4: one tile block per 4-wave
workgroup; 0: persistent producer / consumer) for a rocprofv3 --pmc pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd.vpr.winograd import wino_fused64
B, H, W = 256, 224, 224
x = torch.randn((B, 64, H, W), device="cuda").contiguous(memory_format=torch.channels_last)
Up = torch.randn((4, 16, 4, 4, 16, 4), device="cuda"); bias = torch.randn(64, device="cuda")
for waves in ("4", "0"):
    os.environ["CSLAM_WF_WAVES"] = waves
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        wino_fused64(x, Up, bias, True, True)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"waves={waves}: {min(ts[1:])*1e3:.3f} ms")
