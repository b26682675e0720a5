#!/usr/bin/env python
"""BASELINE config 5 rehearsed on ONE GPU: the full loop for R robots x P keyframes each -- NetVLAD extract
(VGG-16 fp32, 4096-D) of synthetic place-revisiting 640x480 frames, per-keyframe intra + inter-robot matching in
the reference's causal order through the batched LoopClosureSparseMatching calls, the packed descriptor exchange
between the robots, and the budgeted candidate selection (algebraic connectivity maximisation) on the broker.
In the 8-GPU configuration every robot owns a GPU; here the eight robots take turns on one.

    python tools/perf_c5.py [keyframes_per_robot=12500] [robots=8] [budget=1000] [chunk=250] [drain | drain-async]

drain: every receiver takes a step's remote messages in one call; drain-async: additionally robot r's matching is only ENQUEUED
(LoopClosureSparseMatching.process_local_keyframes_begin) and finished -- read-back, thresholds, candidate edges, wire copy -- after
robot r + 1's extraction has been enqueued, so the host's share of the matching hides under the next extraction on the GPU.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    CH = int(sys.argv[4]) if len(sys.argv) > 4 else 250
    DRAIN = len(sys.argv) > 5 and sys.argv[5].startswith("drain")     # receivers take a step's remote messages in one call
    ASYNC = len(sys.argv) > 5 and sys.argv[5] == "drain-async"
    import torch
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    from cslam_amd.vpr.netvlad import NetVLAD
    from cslam_amd.vpr import heads
    from cslam_amd.wire import PackedDescriptorBuffer

    dev = torch.device("cuda")
    nv = NetVLAD({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376, "frontend.netvlad.pca_dim": 4096}, None)
    g = torch.Generator(device=dev).manual_seed(5)
    n_places = 2000
    # a "place" is a smooth random colour field (a 12x16 grid blown up to 480x640) so that places differ at every
    # scale the network sees; a visit adds +-6 grey levels of pixel noise
    coarse = torch.rand((n_places, 3, 12, 16), generator=g, device=dev)
    places = (torch.nn.functional.interpolate(coarse, size=(480, 640), mode="bilinear", align_corners=False) * 255.0)
    places = places.permute(0, 2, 3, 1).contiguous().to(torch.uint8)
    del coarse
    params = lambda r: {"robot_id": r, "max_nb_robots": R, "frontend.sensor_type": "stereo",   # noqa: E731
                        "frontend.similarity_threshold": 0.6, "frontend.nb_best_matches": 10,
                        "frontend.intra_loop_min_inbetween_keyframes": 20, "frontend.enable_sparsification": True,
                        "evaluation.enable_sparsification_comparison": False}
    # similarity threshold between "same place, new noise" and "different place" for these seeded random weights
    def views(idx):
        noise = torch.randint(-6, 7, (len(idx), 480, 640, 3), generator=g, device=dev, dtype=torch.int16)
        return (places[idx].to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
    # Centre the projection like the reference's fitted PCA does (`pca.mean_`): random VLAD weights put a large
    # common component into every descriptor (all cosines > 0.99), which is not what trained, whitened descriptors
    # look like and sends most queries down the exact-scan fallback of the certificate.
    with torch.no_grad():
        nv.compute_embeddings_device(views(torch.arange(4, device=dev)))          # builds the trunk runner
        xcal = nv.trunk(heads.preprocess(views(torch.arange(256, 512, device=dev)), 376))
        vcal = nv.pool(xcal.contiguous())
        nv.pca_mean_proj = (vcal @ nv.pca_components.T).mean(dim=0).contiguous()
    cal = torch.arange(64, device=dev)
    d1, d2 = nv.compute_embeddings_device(views(cal)), nv.compute_embeddings_device(views(cal))
    sims = (d1 @ d2.T).cpu().numpy()
    same, diff = np.diag(sims), sims[~np.eye(64, dtype=bool)]
    # a query is compared with thousands of rows, the calibration with 4032 pairs: stay close to the same-place side
    thr = float(max(1.0 - 4.0 * (1.0 - same.min()), (same.min() + diff.max()) / 2))
    print(f"calibration: 1 - similarity: same place {1 - same.max():.2e}..{1 - same.min():.2e}, different places "
          f"{1 - diff.max():.2e}..{1 - diff.min():.2e} -> frontend.similarity_threshold 1 - {1 - thr:.2e}")
    base = params
    params = lambda r: dict(base(r), **{"frontend.similarity_threshold": thr})    # noqa: E731
    lc = [LoopClosureSparseMatching(params(r)) for r in range(R)]
    bufs = [PackedDescriptorBuffer(r) for r in range(R)]
    rng = np.random.default_rng(5)
    # each robot walks its own stretch of places and visits 3 % of its keyframes somewhere random (loop closures)
    walk = [(rng.integers(0, n_places) + np.arange(P) // 8) % n_places for _ in range(R)]
    for r in range(R):
        jump = rng.random(P) < 0.03
        walk[r] = np.where(jump, rng.integers(0, n_places, size=P), walk[r])
    t_ext = t_loc = t_rem = 0.0
    n_intra = n_inter = 0
    nv.compute_embeddings_device(places[:CH])                     # warm-up
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for s in range(0, P, CH):
        m = min(CH, P - s)
        ids = list(range(s, s + m))
        prev = None

        def retire(p):
            nonlocal n_intra, n_inter
            rr, handle, ev, pinned = p
            intra, inter = handle.finish()
            n_intra += sum(k is not None for _, k in intra)
            n_inter += len(inter)
            ev.synchronize()
            bufs[rr].extend(ids, pinned.numpy())
        for r in range(R):
            t0 = time.perf_counter()
            frames = views(torch.from_numpy(walk[r][s:s + m]).to(dev))
            ddev = nv.compute_embeddings_device(frames)
            if ASYNC:
                t1 = time.perf_counter()
                handle = lc[r].process_local_keyframes_begin(ddev, ids)      # add + searches enqueued, no host synchronisation
                pinned = torch.empty(ddev.shape, dtype=ddev.dtype, pin_memory=True)
                pinned.copy_(ddev, non_blocking=True)                        # the copy that goes on the wire
                ev = torch.cuda.Event(); ev.record()
                if prev is not None:
                    retire(prev)                                             # robot r - 1: under robot r's extraction
                prev = (r, handle, ev, pinned)
                t2 = time.perf_counter()
                t_ext += t1 - t0; t_loc += t2 - t1
                continue
            desc = ddev.cpu().numpy()                              # the copy that goes on the wire
            t1 = time.perf_counter()
            intra, inter = lc[r].process_local_keyframes(ddev, ids)    # matching reads the device tensor in place
            n_intra += sum(k is not None for _, k in intra)
            n_inter += len(inter)
            t2 = time.perf_counter()
            bufs[r].extend(ids, desc)
            if not DRAIN:
                for chunk in bufs[r].chunks(s, 10 ** 9):          # one packed message per chunk of new keyframes
                    for o in range(R):
                        if o != r:
                            n_inter += len(lc[o].process_remote_chunk(chunk, s - 1)[0])
                bufs[r].delete_below(s + m)
            t3 = time.perf_counter()
            t_ext += t1 - t0; t_loc += t2 - t1; t_rem += t3 - t2
        if prev is not None:
            t2 = time.perf_counter()
            retire(prev)
            t_loc += time.perf_counter() - t2
        if DRAIN:
            # every receiver drains its queue once per step: the step's messages of its 7 peers in one call (one search)
            t2 = time.perf_counter()
            msgs = {r: bufs[r].chunks(s, 10 ** 9) for r in range(R)}
            for o in range(R):
                queue = [(c, s - 1) for r in range(R) if r != o for c in msgs[r]]
                n_inter += sum(len(mm) for mm, _ in lc[o].process_remote_chunks(queue))
            for r in range(R):
                bufs[r].delete_below(s + m)
            t_rem += time.perf_counter() - t2
    t_front = time.perf_counter() - t_start
    sel = lc[0].candidate_selector
    n_cand = len(sel.candidate_edges)
    t0 = time.perf_counter()
    chosen = lc[0].select_candidates(K, {r: True for r in range(R)})
    t_sel1 = time.perf_counter() - t0
    sel.candidate_edges_to_fixed(list(chosen))
    t0 = time.perf_counter()
    chosen2 = lc[0].select_candidates(K, {r: True for r in range(R)})     # now with fixed inter-robot links: MAC
    t_sel2 = time.perf_counter() - t0
    n = R * P
    print(f"C5 rehearsal on one GPU: {R} robots x {P} keyframes = {n} keyframes, chunk {CH}")
    print(f"  front end {t_front:.1f} s = {n / t_front:.0f} keyframes/s  [frames+extract {t_ext:.1f} s | local add + intra + inter "
          f"{t_loc:.1f} s | exchange to {R - 1} peers {t_rem:.1f} s]; intra closures {n_intra}, inter-robot matches {n_inter}")
    print(f"  broker: {n_cand} candidate edges on robot 0; select_candidates(K={K}) first call {t_sel1:.2f} s "
          f"({len(chosen)} edges, biased greedy until every robot has a fixed link), second call {t_sel2:.2f} s "
          f"({len(chosen2)} edges, MAC over {n} poses, solver {sel._fiedler_solver()[0] if hasattr(sel, '_fiedler_solver') else '?'})")
    print(f"  whole loop {t_front + t_sel1 + t_sel2:.1f} s -> {n / (t_front + t_sel1 + t_sel2):.0f} keyframes/s")


if __name__ == "__main__":
    main()
