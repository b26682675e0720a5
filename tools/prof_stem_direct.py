#!/usr/bin/env python
"""Where a workgroup of the direct stem kernel (csrc/conv_stem_direct_h.hip) spends its cycles: s_memtime per phase of wave 0 of
workgroup 0 (measurement build: `make abl`, CSLAM_HIP_LIB=cslam_amd/libcslam_hip_abl.so), VGG-16's first two layers at the
256-frame chunk.
    CSLAM_HIP_LIB=cslam_amd/libcslam_hip_abl.so python tools/prof_stem_direct.py [frames=256]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch import nn  # noqa: E402

from cslam_amd import _lib  # noqa: E402
from cslam_amd.vpr import winograd as wg  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    lib = _lib.load()
    buf = torch.zeros(8, dtype=torch.int64, device="cuda")
    assert lib.cslam_debug_sd_prof_dev(C.c_void_p(buf.data_ptr())) == 0
    torch.manual_seed(0)
    seq = nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(),
                        nn.MaxPool2d(2, 2)).cuda().eval()
    x = torch.rand((B, 3, 224, 224), device="cuda") * 4.8 - 2.2
    tr = wg.WinogradTrunk(seq, 64, 4, fused64=True)
    assert tr.steps[0].Wr is not None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tr(x)
    for _ in range(3):
        buf.zero_()
        torch.cuda.synchronize()
        e0.record()
        tr(x)
        e1.record()
        torch.cuda.synchronize()
        h = [int(v) for v in buf.cpu().numpy()]
        n = max(h[4], 1)
        print(f"{e0.elapsed_time(e1):.3f} ms per launch (absmax pass included); wave 0 of workgroup 0: {n} blocks, per block: six columns + next "
              f"first layer {h[0] / n:.0f}, image + epilogue {h[1] / n:.0f}, barrier {h[2] / n:.0f}, sum {sum(h[:3]) / n:.0f} cycles")
    lib.cslam_debug_sd_prof_dev(None)


if __name__ == "__main__":
    main()
