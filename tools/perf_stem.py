#!/usr/bin/env python
"""VGG-16 conv1_1 + conv1_2 (+ pool) at the 256-frame chunk: the direct stem kernel (csrc/conv_stem_direct_h.hip) against the
F(4x4) stem kernel (first layer folded into the one-kernel convolution, csrc/wino_fused_h.hip STEM) and the two separate kernels,
interleaved rounds, median / minimum.
    python tools/perf_stem.py [frames=256] [rounds=5]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch import nn  # noqa: E402

from cslam_amd.vpr.winograd import WinogradTrunk  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    torch.manual_seed(0)
    seq = nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(),
                        nn.MaxPool2d(2, 2)).cuda().eval()
    x = torch.rand((B, 3, 224, 224), device="cuda") * 4.8 - 2.2
    direct = WinogradTrunk(seq, 64, 4, fused64=True)
    assert direct.steps[0].Wr is not None
    stem = WinogradTrunk(seq, 64, 4, fused64=True, forms={"stem_direct": False})
    assert stem.steps[0].Wr is None
    apart = WinogradTrunk(seq, 64, 4, fused64=True, forms={"wino_stem": False})
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def t(fn, n=3):
        e0.record()
        for _ in range(n):
            fn(x)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    ys, ya, yd = stem(x), apart(x), direct(x)
    print("stem vs separate kernels: max |diff| / max |y| = %.1e" % float((ys - ya).abs().max() / ya.abs().max()))
    print("direct stem vs separate kernels: max |diff| / max |y| = %.1e" % float((yd - ya).abs().max() / ya.abs().max()))
    ts, ta, td = [], [], []
    for _ in range(rounds):
        ta.append(t(apart)); ts.append(t(stem)); td.append(t(direct))
    for tag, v in (("conv1_1 + conv1_2 as two kernels", ta), ("F(4x4) stem kernel", ts), ("direct stem kernel", td)):
        print(f"{tag:34s}: median {statistics.median(v):.3f} ms, min {min(v):.3f} ms per {B} frames")


if __name__ == "__main__":
    main()
