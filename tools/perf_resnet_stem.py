"""ResNet's stem (7 x 7 / 2 convolution + ReLU + MaxPool2d(3, 2, 1), csrc/conv_igemm.hip: conv_stem_pool_patch_kernel) on B frames of
224 x 224: python tools/perf_resnet_stem.py [B].  With the measurement build (CSLAM_HIP_LIB=cslam_amd/libcslam_hip_abl.so) CSLAM_SP_DBG
selects the timing-only ablations (1 no pooling phase, 2 no K loop, 4 no patch split, 8 no patch requests, 16 / 32 the atomic maxima)."""
import os
import sys
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr import winograd as wg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ws = wg._Workspace()
torch.zeros(1 << 28, device="cuda").sum().item()
x = torch.randn((B, 3, 224, 224), device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn((64, 3, 7, 7), device="cuda") / 12
bias = torch.randn(64, device="cuda") * 0.1
Wg = wg.igemm_pair_weights(w)
slot = torch.full((1,), float(x.abs().max()), device="cuda")


def run():
    return wg.conv_igemm(ws, x, Wg, bias, (7, 7), 2, 3, True, amax_in=slot, pool=True)


run()
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * 3 * B * 112 * 112 * 64 * 224
    print(f"stem + pool, {B} frames, CSLAM_SP_DBG={os.environ.get('CSLAM_SP_DBG', '0')}: {ms:.3f} ms  ({fl / ms / 1e9:.0f} TF issued with K = 224; zero fill included)", flush=True)
