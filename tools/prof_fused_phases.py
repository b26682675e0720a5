#!/usr/bin/env python
"""Where a workgroup of the one-kernel convolution spends its cycles (CSLAM_WFH_PROF=1: s_memtime per phase, waves 0 and 4 of
workgroup 0), conv1_2's shape at the 256-frame chunk, for the plain / frequency-split (XS) / stem forms.
    python tools/prof_fused_phases.py [frames=256]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CSLAM_WFH_PROF"] = "1"
import torch  # noqa: E402
from torch import nn  # noqa: E402

from cslam_amd import _lib  # noqa: E402
from cslam_amd.vpr import winograd as wg  # noqa: E402

NAMES = ("transform", "barrier 1", "matrix loop", "prefetch/stem", "output transform", "barrier 2")


def report(tag, buf, ms):
    h = buf.cpu().numpy().reshape(2, 8)
    for w in (0, 1):
        q = max(int(h[w, 6]), 1)
        tot = sum(int(h[w, i]) for i in range(6))
        parts = ", ".join(f"{NAMES[i]} {int(h[w, i]) / q:7.0f}" for i in range(6))
        print(f"{tag:22s} wave {4 * w}: {q} quarters, {tot / q:7.0f} cycles per quarter ({ms:.3f} ms per launch) | {parts}")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    lib = _lib.load()
    buf = torch.zeros(16, dtype=torch.int64, device="cuda")
    _lib.check(lib.cslam_debug_wfh_prof_dev(C.c_void_p(buf.data_ptr())))
    torch.manual_seed(0)
    seq = nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(),
                        nn.MaxPool2d(2, 2)).cuda().eval()
    x = torch.rand((B, 3, 224, 224), device="cuda") * 4.8 - 2.2
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for xs in ("0",):
        os.environ["CSLAM_WFH_XS"] = xs
        for stem in ("0", "1"):
            os.environ["CSLAM_WINO_STEM"] = stem
            tr = wg.WinogradTrunk(seq, 64, 4, fused64=True)
            tr(x)
            buf.zero_()
            torch.cuda.synchronize()
            e0.record()
            tr(x)
            e1.record()
            torch.cuda.synchronize()
            report(f"XS={xs} stem={stem}", buf, e0.elapsed_time(e1))
    # conv2_1's shape: 64 -> 128 channels on the pooled map, one block per iteration
    seq2 = nn.Sequential(nn.Conv2d(64, 128, 3, padding=1), nn.ReLU()).cuda().eval()
    x2 = torch.relu(torch.randn((B, 64, 112, 112), device="cuda")).contiguous(memory_format=torch.channels_last)
    tr = wg.WinogradTrunk(seq2, 64, 4, fused64=True)
    tr(x2)
    buf.zero_()
    torch.cuda.synchronize()
    e0.record()
    tr(x2)
    e1.record()
    torch.cuda.synchronize()
    report("conv2_1 (64 -> 128)", buf, e0.elapsed_time(e1))
    _lib.check(lib.cslam_debug_wfh_prof_dev(None))


if __name__ == "__main__":
    main()
