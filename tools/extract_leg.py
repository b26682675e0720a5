#!/usr/bin/env python
"""The extract leg alone (NetVLAD VGG-16, chunks of 256 resident frames) for rocprofv3 runs:
2 warm-up passes (MIOpen find mode, workspace growth), then `--iters` steady-state passes.  Every pass
starts with `preprocess_fused_kernel`, which tools/kernel_trace_summary.py uses to cut the trace."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd.vpr.netvlad import NetVLAD

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--backbone-conv", default="winograd")
a = ap.parse_args()
torch.backends.cudnn.benchmark = True
nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096,
              "frontend.backbone_conv": a.backbone_conv}, None)
fr = torch.randint(0, 256, (a.batch, 480, 640, 3), device="cuda", dtype=torch.uint8)
import time
for _ in range(2):
    nv.compute_embeddings_device(fr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    nv.compute_embeddings_device(fr)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
print("extract leg: %.2f ms per %d frames = %.0f frames/s" % (dt * 1e3, a.batch, a.batch / dt))
