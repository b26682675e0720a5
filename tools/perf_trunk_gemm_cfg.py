"""Whole 256-frame passes of the VGG-16 trunk with the pair products' shape forced (CSLAM_WGEMM_CFG; read per call), interleaved:
2 = 256 x 128 ring of 3 with the waves as 2 x 4, 5 = the same with the waves as 4 x 2 (the default)."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cslam_amd.vpr import heads  # noqa: E402
from cslam_amd.vpr.netvlad import NetVLAD  # noqa: E402
from cslam_amd.vpr.winograd import WinogradTrunk  # noqa: E402

nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096}, None)
B = 256
fr = torch.randint(0, 256, (B, 480, 640, 3), device="cuda", dtype=torch.uint8)
x = heads.preprocess(fr, 376)
tr = WinogradTrunk(nv.encoder, min_in_channels=64, tile=4)
tr.input_bound = heads.normalised_image_bound()
cfgs = sys.argv[1:] or ["2", "5"]
res = {c: [] for c in cfgs}
for rnd in range(8):
    for c in (cfgs if rnd % 2 == 0 else cfgs[::-1]):
        os.environ["CSLAM_WGEMM_CFG"] = c
        tr(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            tr(x)
        torch.cuda.synchronize()
        res[c].append((time.perf_counter() - t0) / 4 * 1e3)
for c in cfgs:
    print("CSLAM_WGEMM_CFG=%s: %.3f ms per 256-frame trunk pass (median of %s)" % (c, statistics.median(res[c]), ["%.2f" % v for v in res[c]]))
