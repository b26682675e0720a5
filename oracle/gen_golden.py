#!/usr/bin/env python
"""oracle/gen_golden.py -- TEST INFRASTRUCTURE.  Generates tests/golden/*.npz.

Runs ONLY in the build container (it imports the real reference from
/root/reference); the fixtures it writes are data (inputs / seeds + the
reference's outputs) and travel to the GPU box, the reference never does.

    python oracle/gen_golden.py [nns] [seq] [heads] [mac] [acm]

Stubs: the reference imports `numba` (never used: cslam/mac/utils.py:9-10),
`torchvision`, `ament_index_python` at module top; none is installed here, so
no-op stand-in modules are injected into sys.modules for the import only.  They
provide no arithmetic -- everything recorded below is computed by the reference's
own code plus numpy/scipy/torch/PIL/sklearn/networkx as installed.
"""
import hashlib
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_stubs():
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    sys.modules.setdefault("numba", nb)
    for name in ("torchvision", "torchvision.transforms", "torchvision.datasets",
                 "torchvision.models", "ament_index_python", "ament_index_python.packages"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["ament_index_python.packages"].get_package_share_directory = lambda p: "/nonexistent"
    if REF not in sys.path:
        sys.path.insert(0, REF)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def unit_rows(rng, n, d):
    """The synthetic-descriptor recipe of BASELINE.md section 3 / SURVEY 8(d)."""
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return x


# ---------------------------------------------------------------- G1: NNS ----
def gen_nns():
    from cslam.nns_matching import NearestNeighborsMatching

    cases = {}
    meta = []

    def run_case(name, bank, queries, k, store_inputs):
        nn = NearestNeighborsMatching()
        for i in range(bank.shape[0]):
            nn.add_item(bank[i], int(i))
        idx = np.full((queries.shape[0], k), -1, dtype=np.int64)
        sims = np.full((queries.shape[0], k), np.nan, dtype=np.float64)
        cnt = np.zeros(queries.shape[0], dtype=np.int32)
        min_gap = np.inf
        for j in range(queries.shape[0]):
            items, s = nn.search(queries[j], k)
            cnt[j] = len(items)
            idx[j, :len(items)] = items
            sims[j, :len(items)] = s
            # tie-freeness of the fixture: smallest gap between consecutive scores
            # among the top k+1 (so the k-th boundary is covered too)
            items2, s2 = nn.search(queries[j], k + 1)
            if len(s2) > 1:
                min_gap = min(min_gap, float(np.min(-np.diff(np.asarray(s2, dtype=np.float64)))))
        assert np.array_equal(nn.data[:nn.n], bank.astype(np.float32))
        cases[name + "/idx"] = idx
        cases[name + "/sims"] = sims
        cases[name + "/cnt"] = cnt
        cases[name + "/k"] = np.int64(k)
        cases[name + "/bank_sha"] = np.array(sha(bank.astype(np.float32)))
        cases[name + "/q_sha"] = np.array(sha(queries))
        if store_inputs:
            cases[name + "/bank"] = bank.astype(np.float32)
            cases[name + "/queries"] = queries
        meta.append((name, bank.shape, queries.dtype.name, k, min_gap))
        print(f"  {name}: bank {bank.shape} q {queries.shape} {queries.dtype} k={k} min_gap={min_gap:.3e}")
        return min_gap

    # C1 = BASELINE config 1: 1k x 4096, top-5.  Inputs regenerated from seeds in the
    # tests (sha256 recorded); 64 of the 1000 queries go through the reference here
    # (12 ms / query) -- the full 1000-query C1 timing is recorded by `time_c1`.
    for seed in range(2):
        bank = unit_rows(np.random.default_rng(1234 + seed), 1000, 4096)
        q32 = unit_rows(np.random.default_rng(4321 + seed), 64, 4096)
        g = run_case(f"c1_s{seed}_f32", bank, q32, 5, False)
        assert g > 1e-6, "fixture not tie-free"
        g = run_case(f"c1_s{seed}_f64", bank, q32.astype(np.float64), 5, False)
        assert g > 1e-6
    # ragged sizes / dims / k (inputs stored: small)
    for (n, d, k, seed) in [(257, 512, 10, 2), (257, 64, 1, 3), (1, 64, 5, 4), (33, 128, 36, 5),
                            (300, 4096, 10, 7)]:
        bank = unit_rows(np.random.default_rng(100 + seed), n, d)
        q = unit_rows(np.random.default_rng(200 + seed), 16, d)
        g = run_case(f"r_n{n}_d{d}_k{k}_f32", bank, q, k, n * d <= 40000)
        assert g > 1e-6 or n == 1
        qd = np.random.default_rng(300 + seed).standard_normal((16, d))  # genuinely f64, not unit
        g = run_case(f"r_n{n}_d{d}_k{k}_f64", bank, qd, k, n * d <= 40000)
        assert g > 1e-9 or n == 1
    # non-unit-norm bank and queries (per-pair normalisation at search time)
    rng = np.random.default_rng(77)
    bank = (rng.standard_normal((200, 96)) * rng.uniform(0.1, 30.0, size=(200, 1))).astype(np.float32)
    q = (rng.standard_normal((8, 96)) * 5.0).astype(np.float32)
    assert run_case("nonunit_f32", bank, q, 7, True) > 1e-6
    # query identical to a bank row: similarity clips to exactly <= 1
    bank = unit_rows(np.random.default_rng(78), 50, 256)
    q = bank[[3, 17, 49]].copy()
    run_case("selfmatch_f32", bank, q, 3, True)
    run_case("selfmatch_f64", bank, q.astype(np.float64), 3, True)

    # empty bank behaviour (nns_matching.py:52-53, 72-73)
    nn = NearestNeighborsMatching()
    assert nn.search(np.zeros(4, dtype=np.float32), 3) == ([], [])
    assert nn.search_best(np.zeros(4, dtype=np.float32)) == (None, None)

    np.savez_compressed(os.path.join(OUT, "nns_g1.npz"), **cases)
    return meta


# ------------------------------------------- G2: causal multi-robot replay ----
def gen_seq():
    """Replays gdlcd.py:148-174 ordering (intra search -> add -> inter best-1) and
    lcsm.py:56-72 (remote descriptor arrives as a float64 list) on 3 robots."""
    from collections import namedtuple
    from cslam.loop_closure_sparse_matching import LoopClosureSparseMatching

    GlobalDescriptor = namedtuple("GlobalDescriptor", ["keyframe_id", "robot_id", "descriptor"])
    out = {}
    for thr in (0.0, 0.1):
        R, T, D = 3, 120, 128
        rng = np.random.default_rng(9)
        # descriptors with structure: a few shared "places" so matches exceed 0.1
        places = unit_rows(rng, 40, D)
        desc = np.zeros((R, T, D), dtype=np.float32)
        for r in range(R):
            for t in range(T):
                v = places[rng.integers(0, 40)] + 0.35 * rng.standard_normal(D).astype(np.float32)
                desc[r, t] = v / np.linalg.norm(v)
        lcsms = []
        for r in range(R):
            params = {"robot_id": r, "max_nb_robots": R, "frontend.sensor_type": "stereo",
                      "frontend.similarity_threshold": thr, "frontend.nb_best_matches": 10,
                      "frontend.intra_loop_min_inbetween_keyframes": 20,
                      "frontend.enable_sparsification": True,
                      "evaluation.enable_sparsification_comparison": False}
            lcsms.append(LoopClosureSparseMatching(params))
        intra, inter_local, inter_remote = [], [], []
        for t in range(T):
            for r in range(R):
                emb = desc[r, t]
                kf, kfs = lcsms[r].match_local_loop_closures(emb, t)
                intra.append((r, t, -1 if kf is None else kf))
                for m in lcsms[r].add_local_global_descriptor(emb, t):
                    inter_local.append((r, m.robot0_id, m.robot0_keyframe_id, m.robot1_id,
                                        m.robot1_keyframe_id, m.weight))
                # wire format: embedding.tolist() -> float32 msg -> np.asarray (f64)  gdlcd.py:167, lcsm.py:63
                msg = GlobalDescriptor(t, r, emb.tolist())
                for o in range(R):
                    if o != r:
                        m = lcsms[o].add_other_robot_global_descriptor(msg)
                        if m is not None:
                            inter_remote.append((o, m.robot0_id, m.robot0_keyframe_id, m.robot1_id,
                                                 m.robot1_keyframe_id, m.weight))
        tag = f"thr{thr}"
        out[tag + "/desc"] = desc
        out[tag + "/intra"] = np.array(intra, dtype=np.int64)
        out[tag + "/inter_local"] = np.array(inter_local, dtype=np.float64).reshape(-1, 6)
        out[tag + "/inter_remote"] = np.array(inter_remote, dtype=np.float64).reshape(-1, 6)
        for r in range(R):
            keys = sorted(lcsms[r].candidate_selector.candidate_edges.keys())
            w = [lcsms[r].candidate_selector.candidate_edges[k].weight for k in keys]
            out[tag + f"/cand_keys_r{r}"] = np.array(keys, dtype=np.int64).reshape(-1, 4)
            out[tag + f"/cand_w_r{r}"] = np.array(w, dtype=np.float64)
        print(f"  seq thr={thr}: intra hits {sum(1 for x in intra if x[2] >= 0)}, "
              f"inter_local {len(inter_local)}, inter_remote {len(inter_remote)}")
    np.savez_compressed(os.path.join(OUT, "seq_g2.npz"), **out)


def time_c1():
    """Times the REAL reference on BASELINE config 1 (1k x 4096, top-5), 200 queries."""
    import time
    from cslam.nns_matching import NearestNeighborsMatching
    bank = unit_rows(np.random.default_rng(1234), 1000, 4096)
    q = unit_rows(np.random.default_rng(4321), 200, 4096)
    nn = NearestNeighborsMatching()
    for i in range(1000):
        nn.add_item(bank[i], i)
    t0 = time.perf_counter()
    for j in range(200):
        nn.search(q[j], 5)
    dt = time.perf_counter() - t0
    print(f"  reference C1: {dt / 200 * 1e3:.2f} ms/query, {200 / dt:.1f} keyframes/s (1 core)")


if __name__ == "__main__":
    _install_stubs()
    os.makedirs(OUT, exist_ok=True)
    what = sys.argv[1:] or ["nns", "seq", "heads", "mac", "acm"]
    if "nns" in what:
        print("G1 nns"); gen_nns()
    if "seq" in what:
        print("G2 seq"); gen_seq()
    if "time" in what:
        time_c1()
    if "heads" in what:
        from gen_golden_heads import gen_heads
        print("G3-G6 heads"); gen_heads(OUT)
    if "mac" in what or "acm" in what:
        from gen_golden_mac import gen_mac
        print("G7/G8 mac+acm"); gen_mac(OUT)
