"""Whole NetVLAD extract passes (256 frames), interleaved A/B of trunk variants (vpr/winograd.py TRUNK_FORMS, key=value[,key=value]):
    python tools/perf_trunk_ab.py "conv_direct=1" "conv_direct=2" "conv_direct=0" """
import os, sys, statistics
import torch
sys.path.insert(0, ".")
from cslam_amd.vpr.netvlad import NetVLAD
from cslam_amd import synthetic

variants = sys.argv[1:] or ["conv_direct=1", "conv_direct=0"]
frames = torch.from_numpy(synthetic.frames(0, 256)).cuda()
runners = []
for v in variants:
    forms = {}
    for kv in v.split(","):
        k, val = kv.split("=")
        forms[k] = {"True": True, "False": False, "None": None}.get(val, int(val) if val.lstrip("-").isdigit() else val)
    ex = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096,
                  "frontend.random_seed": 0, "frontend.backbone_conv": "winograd", "frontend.trunk_forms": forms}, None)
    ex.compute_embeddings_device(frames)
    runners.append(ex)
torch.cuda.synchronize()
res = [[] for _ in runners]
outs = [None] * len(runners)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rnd in range(7):
    for i, ex in enumerate(runners):
        e0.record()
        for _ in range(2):
            outs[i] = ex.compute_embeddings_device(frames)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            res[i].append(e0.elapsed_time(e1) / 2)
for i, v in enumerate(variants):
    d = float((outs[i] - outs[0]).abs().max())
    print(f"{v:40s}: {statistics.median(res[i]):7.3f} ms per 256-frame pass (min {min(res[i]):.3f}); max |descriptor - first variant's| {d:.2e}", flush=True)
