"""NetVLAD extract of 256 frames with the trunk's Winograd layers run in sub-batches (CSLAM_WINO_SUB_MB): does keeping V and M
in the memory-side cache between the three kernels of a layer pay?  Also the descriptors must not depend on the split."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cslam_amd.vpr.netvlad import NetVLAD  # noqa: E402

nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096}, None)
fr = torch.randint(0, 256, (256, 480, 640, 3), device="cuda", dtype=torch.uint8)
ref = None
for mb in sys.argv[1:] or ["0", "192", "128", "96", "64", "48", "32", "16", "0"]:
    os.environ["CSLAM_WINO_SUB_MB"] = mb
    for _ in range(2):
        d = nv.compute_embeddings_device(fr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        d = nv.compute_embeddings_device(fr)
    e1.record()
    torch.cuda.synchronize()
    if ref is None:
        ref = d.clone()
    print("CSLAM_WINO_SUB_MB=%-4s  %.2f ms per 256 frames   identical to the unsplit pass: %s" % (mb, e0.elapsed_time(e1) / 4, bool(torch.equal(d, ref))), flush=True)
