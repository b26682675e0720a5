#!/usr/bin/env python
"""Online (one keyframe at a time, the ROS deployment) latency of the path on the GPU box:
compute_embedding (B = 1) stage by stage, then the per-keyframe matcher calls on a 100k bank.

    python tools/perf_online.py [--bank-rows 100000]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, n, sync):
    for _ in range(3):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bank-rows", type=int, default=100000)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    import torch
    from cslam_amd.vpr.netvlad import NetVLAD
    from cslam_amd.vpr.cosplace import CosPlace
    from cslam_amd.vpr import heads
    from cslam_amd.nns_matching import NearestNeighborsMatching

    sync = torch.cuda.synchronize
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    params = {"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096,
              "frontend.cosplace.descriptor_dim": 512, "frontend.cosplace.backbone": "resnet18"}
    nv = NetVLAD(params, None)
    t_all = timeit(lambda: nv.compute_embedding(frame), a.iters, sync)
    dframe = torch.from_numpy(frame).cuda().unsqueeze(0)
    t_h2d = timeit(lambda: torch.from_numpy(frame).cuda(), a.iters, sync)
    t_pre = timeit(lambda: heads.preprocess(dframe, 376), a.iters, sync)
    x = heads.preprocess(dframe, 376)
    with torch.no_grad():
        from cslam_amd.vpr.winograd import WinogradTrunk
        wt = WinogradTrunk(nv.encoder, 64, 2)
        t_enc = timeit(lambda: wt(x), a.iters, sync)
        f = nv.encoder(x)
        t_vlad = timeit(lambda: nv.pool(f), a.iters, sync)
        v = nv.pool(f)
        t_pca = timeit(lambda: heads.pca_project(v, nv.pca_components, nv.pca_mean_proj, nv.pca_inv_scale), a.iters, sync)
    print(f"NetVLAD compute_embedding B=1: {t_all * 1e3:.3f} ms/keyframe  "
          f"[H2D {t_h2d * 1e6:.0f} us | preprocess {t_pre * 1e6:.0f} us | VGG-16 {t_enc * 1e6:.0f} us | "
          f"VLAD {t_vlad * 1e6:.0f} us | PCA {t_pca * 1e6:.0f} us]")
    cp = CosPlace(params, None)
    t_cp = timeit(lambda: cp.compute_embedding(frame), a.iters, sync)
    print(f"CosPlace (ResNet-18, 512-D) compute_embedding B=1: {t_cp * 1e3:.3f} ms/keyframe")

    bank = rng.standard_normal((a.bank_rows, 4096)).astype(np.float32)
    bank /= np.linalg.norm(bank, axis=1, keepdims=True)
    m = NearestNeighborsMatching()
    m.add_items(bank, range(a.bank_rows))
    q = bank[123] + 0.01 * rng.standard_normal(4096).astype(np.float32)
    t_s = timeit(lambda: m.search(q, 5), a.iters, sync)
    t_b = timeit(lambda: m.search_best(q.astype(np.float64)), a.iters, sync)
    t_add = timeit(lambda: m.add_item(q, 0), a.iters, sync)
    print(f"NearestNeighborsMatching on {a.bank_rows}x4096 (host API, includes PCIe + launches): "
          f"search(k=5) {t_s * 1e6:.0f} us, search_best(float64 query) {t_b * 1e6:.0f} us, add_item {t_add * 1e6:.0f} us")
    nvg = NetVLAD(dict(params, **{"frontend.hip_graph": True}), None)     # measured last: leaves a one-off stall behind
    t_graph = timeit(lambda: nvg.compute_embedding(frame), a.iters, sync)
    print(f"NetVLAD compute_embedding replayed from a captured HIP graph (frontend.hip_graph: true): {t_graph * 1e3:.3f} ms/keyframe")
    print(f"one keyframe end to end (embed + intra search + add + 1 inter search): "
          f"{(t_all + t_s + t_add + t_b) * 1e3:.3f} ms -> {1.0 / (t_all + t_s + t_add + t_b):.0f} keyframes/s per stream")


if __name__ == "__main__":
    main()
