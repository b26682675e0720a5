"""conv2_2's two kernels back to back for tools/power_trace.py: python tools/power_conv2_2_target.py [h|r2] [seconds]"""
import sys
import time
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr import winograd as wg

which = sys.argv[1] if len(sys.argv) > 1 else "r2"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
x = torch.randn((256, 128, 112, 112), device="cuda").relu().contiguous(memory_format=torch.channels_last)
w = torch.randn((128, 128, 3, 3), device="cuda") / 34
bias = torch.randn(128, device="cuda") * 0.1
Wd, Wr2 = wg.direct_pair_weights(w), wg.direct_r2_pair_weights(w)
slot = torch.full((1,), float(x.abs().max()), device="cuda")
fn = (lambda: wg.conv3x3_direct_h(x, Wd, bias, True, True, slot)) if which == "h" else (lambda: wg.conv3x3_direct_r2(x, Wr2, bias, True, True, slot))
fn()
torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    n += 50
print(f"{which}: {n} launches, {(time.time() - t0) / n * 1e3:.3f} ms per launch")
