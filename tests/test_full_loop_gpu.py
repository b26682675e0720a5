"""GPU test of the whole hot path in the reference's call order (-m gpu), BASELINE config 5 in small:
synthetic keyframes -> CosPlace extract (HIP heads) -> LoopClosureSparseMatching (HBM banks, HIP
search) -> candidate edges -> AlgebraicConnectivityMaximization.select_candidates (MAC), for 3
robots; compared with the same loop driven by the CPU oracle for the matching part."""
from collections import namedtuple

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GlobalDescriptor = namedtuple("GlobalDescriptor", ["keyframe_id", "robot_id", "descriptor"])


def test_extract_match_select_three_robots():
    import torch
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    from cslam_amd.algebraic_connectivity_maximization import EdgeInterRobot
    from cslam_amd.vpr.cosplace import CosPlace
    from oracle import pyoracle
    R, T = 3, 60
    cp = CosPlace({"frontend.nn_checkpoint": "random", "frontend.image_crop_size": 376,
                   "frontend.cosplace.descriptor_dim": 64, "frontend.cosplace.backbone": "resnet18"}, None)
    # robots revisit a shared set of 25 "places": frame = place image + noise
    rng = np.random.default_rng(0)
    places = rng.integers(0, 256, size=(25, 480, 640, 3), dtype=np.uint8)
    desc = np.zeros((R, T, 64), dtype=np.float32)
    for r in range(R):
        idx = rng.integers(0, 25, size=T)
        frames = np.clip(places[idx].astype(np.int16) + rng.integers(-6, 7, size=(T, 480, 640, 3)), 0, 255).astype(np.uint8)
        desc[r] = cp.compute_embeddings_device(torch.from_numpy(frames).cuda()).cpu().numpy()
    params = lambda r: {"robot_id": r, "max_nb_robots": R, "frontend.sensor_type": "stereo",
                        "frontend.similarity_threshold": 0.5, "frontend.nb_best_matches": 10,
                        "frontend.intra_loop_min_inbetween_keyframes": 5,
                        "frontend.enable_sparsification": True,
                        "evaluation.enable_sparsification_comparison": False}
    lc = [LoopClosureSparseMatching(params(r)) for r in range(R)]
    expected = [dict() for _ in range(R)]            # oracle-driven candidate edges per robot
    banks = [np.zeros((0, 64), dtype=np.float32) for _ in range(R)]   # what robot r knows of each robot
    known = [[np.zeros((0, 64), dtype=np.float32) for _ in range(R)] for _ in range(R)]

    def oracle_best(bank, q):
        if bank.shape[0] == 0:
            return None, None
        i, s, _ = pyoracle.nns_search(bank, q[None, :], 1)
        return int(i[0, 0]), float(s[0, 0])

    def add_expected(r, e):
        sel = lc[r].candidate_selector
        key = (e.robot0_id, e.robot0_keyframe_id, e.robot1_id, e.robot1_keyframe_id)
        nkey = sel.edge_key(e)
        if key in expected[r] and not (e.weight > expected[r][key].weight):
            return
        expected[r][nkey] = e

    for t in range(T):
        for r in range(R):
            emb = desc[r, t]
            lc[r].match_local_loop_closures(emb, t)
            lc[r].add_local_global_descriptor(emb, t)
            for o in range(R):                                   # oracle twin of lcsm.py:45-53
                if o != r:
                    kf, s = oracle_best(known[r][o], emb)
                    if kf is not None and s >= 0.5:
                        add_expected(r, EdgeInterRobot(r, t, o, kf, s))
            known[r][r] = np.vstack([known[r][r], emb[None]])
            msg = GlobalDescriptor(t, r, emb.tolist())
            for o in range(R):
                if o != r:
                    lc[o].add_other_robot_global_descriptor(msg)
                    kf, s = oracle_best(known[o][o], emb.astype(np.float64))   # lcsm.py:66
                    known[o][r] = np.vstack([known[o][r], emb[None]])
                    if kf is not None and s >= 0.5:
                        add_expected(o, EdgeInterRobot(o, kf, r, t, s))
    for r in range(R):
        got = lc[r].candidate_selector.candidate_edges
        assert sorted(got) == sorted(expected[r]), f"robot {r}: candidate edge sets differ"
        assert len(got) > 10
        for kx in got:
            assert abs(got[kx].weight - expected[r][kx].weight) < 1e-6
    # budgeted selection on the broker robot: first call has no fixed inter-robot links (biased greedy),
    # after fixing them MAC runs (reference acm.py:513-530)
    sel0 = lc[0].select_candidates(6, {0: True, 1: True, 2: True})
    assert len(sel0) == 6 and len(set(sel0)) == 6
    lc[0].candidate_selector.candidate_edges_to_fixed(list(sel0))
    if all(lc[0].candidate_selector.initial_fixed_edge_exists[r] for r in range(R)):
        sel1 = lc[0].select_candidates(5, {0: True, 1: True, 2: True})
        assert len(sel1) == 5 and all(e not in sel0 for e in sel1)


def test_reference_checkpoint_layouts_load(tmp_path):
    """Checkpoints named like the reference's (netvlad.py:187-197 'state_dict' with encoder.*/pool.*,
    cosplace.py:60-67 raw state_dict with backbone.*/aggregation.*) load into the HIP pipeline."""
    import pickle
    import torch
    from sklearn.decomposition import PCA
    from cslam_amd.vpr.backbones import get_backbone
    from cslam_amd.vpr.cosplace import CosPlace
    from cslam_amd.vpr.netvlad import NetVLAD
    torch.manual_seed(0)
    vgg, _ = get_backbone("vgg16")
    sd = {"encoder.module." + k: v for k, v in vgg.state_dict().items()}       # DataParallel naming
    sd["pool.module.conv.weight"] = torch.randn(64, 512, 1, 1)
    sd["pool.module.centroids"] = torch.rand(64, 512)
    ck = tmp_path / "netvlad.pth.tar"
    torch.save({"epoch": 3, "best_score": 0.5, "state_dict": sd}, ck)
    pca = PCA(n_components=16).fit(np.random.default_rng(0).standard_normal((40, 32768)).astype(np.float32))
    pk = tmp_path / "pca.pkl"
    pickle.dump(pca, open(pk, "wb"))
    nv = NetVLAD({"frontend.nn_checkpoint": str(ck), "frontend.netvlad.pca_checkpoint": str(pk),
                  "frontend.image_crop_size": 376}, None)
    img = np.random.default_rng(1).integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    e = nv.compute_embedding(img)
    assert e.shape == (16,) and abs(np.linalg.norm(e) - 1) < 1e-5
    assert torch.equal(nv.pool.centroids.cpu(), sd["pool.module.centroids"])
    r18, _ = get_backbone("resnet18")
    sd = {"backbone." + k: v for k, v in r18.state_dict().items()}
    sd["aggregation.1.p"] = torch.tensor([2.5])
    sd["aggregation.3.weight"] = torch.randn(64, 512) / 20
    sd["aggregation.3.bias"] = torch.zeros(64)
    ck2 = tmp_path / "cosplace.pth"
    torch.save(sd, ck2)
    cp = CosPlace({"frontend.nn_checkpoint": str(ck2), "frontend.image_crop_size": 376,
                   "frontend.cosplace.descriptor_dim": 64, "frontend.cosplace.backbone": "resnet18"}, None)
    assert abs(cp.model.gem_p - 2.5) < 1e-6
    e = cp.compute_embedding(img)
    assert e.shape == (64,) and abs(np.linalg.norm(e) - 1) < 1e-5
