#!/usr/bin/env python
"""Do two 256-frame passes of the trunk on two HIP streams overlap the HBM-bound kernels of one (transforms, the 128 / 256-channel
products) with the MFMA- / latency-bound kernels of the other (512-channel products, one-kernel convolutions)?  Round 1 measured
half batches on two streams with the library-GEMM trunk (slower); this is today's trunk with whole chunks per stream."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cslam_amd.vpr import heads  # noqa: E402
from cslam_amd.vpr.netvlad import NetVLAD  # noqa: E402
from cslam_amd.vpr.winograd import WinogradTrunk  # noqa: E402

nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096}, None)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fr = torch.randint(0, 256, (B, 480, 640, 3), device="cuda", dtype=torch.uint8)
x = heads.preprocess(fr, 376)


def trunk():
    t = WinogradTrunk(nv.encoder, min_in_channels=64, tile=4)
    t.input_bound = heads.normalised_image_bound()
    return t


def bench(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


t1 = trunk()
print("one stream, %d frames per pass: %.2f ms per pass" % (B, 1e3 * bench(lambda: t1(x))), flush=True)
for ns in (2, 3):
    trunks = [trunk() for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]

    def run():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for tr, s in zip(trunks, streams):
            with torch.cuda.stream(s):
                tr(x)
        for s in streams:
            cur.wait_stream(s)
    print("%d streams, %d frames per pass each: %.2f ms per pass" % (ns, B, 1e3 * bench(run) / ns), flush=True)

# free-running lanes (no join between passes), with and without half a pass of offset between them
for ns, offset in ((2, False), (2, True), (3, True)):
    trunks = [trunk() for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    xh = x[: B // 2]
    n = 8

    def run_free():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for i, (tr, s) in enumerate(zip(trunks, streams)):
            with torch.cuda.stream(s):
                if offset and i:
                    for _ in range(i):
                        tr(xh[: (B // 2) // (ns - 1)] if ns > 2 else xh)      # offset work (counted in the time, not in the frames)
                for _ in range(n):
                    tr(x)
        for s in streams:
            cur.wait_stream(s)
    run_free()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_free()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d free-running lanes%s, %d passes of %d frames each: %.2f ms per pass (offset work included in the time)"
          % (ns, " offset by part of a pass" if offset else "", n, B, 1e3 * dt / (ns * n)), flush=True)
