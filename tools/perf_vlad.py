"""Time the channels-last VLAD aggregation (cslam_vlad_aggregate_nhwc_dev) at NetVLAD's shape, B frames per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cslam_amd.vpr import heads  # noqa: E402


def run(tag, B=256, reps=30):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand((B, 512, 14, 14), generator=g, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn((64, 512), generator=g, device="cuda")
    b = torch.randn(64, generator=g, device="cuda")
    c = torch.rand((64, 512), generator=g, device="cuda")
    for _ in range(3):
        heads.vlad_aggregate(x, w, b, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        heads.vlad_aggregate(x, w, b, c)
    e1.record()
    torch.cuda.synchronize()
    print("%-28s B=%d  %.1f us per launch" % (tag, B, 1e3 * e0.elapsed_time(e1) / reps))


if __name__ == "__main__":
    for B in (256, 512, 1024):
        run("matrix form (vlad_mfma_kernel)", B=B)
