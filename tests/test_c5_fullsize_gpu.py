"""BASELINE config 5's FRONT END at its real size on one GPU (-m gpu): 8 robots x 125 000 keyframes = 10^6 4096-D descriptors
streamed through `LoopClosureSparseMatching.process_local_keyframes` (local bank + this robot's copies of the 7 other banks:
cslam/global_descriptor_loop_closure_detection.py:148-174, loop_closure_sparse_matching.py:36-72) and, once per step, every
receiver's drained queue of the other robots' packed descriptor messages through `process_remote_chunks` (gdlcd.py:198-227,
407-422) -- the reference's causal order, 64 banks of up to 125 000 rows resident (~ 205 GB of HBM: the float32 rows and
their fp16 candidate-stage copy).  Oracle-sampled on > 400 keyframes: the intra-robot decision (top-10 + gap + threshold) and
the inter-robot best-1 in both directions; then the candidate edges robot 0 has collected go through
`select_candidates(K = 1000)` over the 10^6-pose graph (acm.py:468-543) with the default HIP solver -- the step
tests/test_fullsize_gpu.py::test_c5_selection_over_one_million_poses checks against the reference's algorithm on a synthetic
candidate set.  (tests/test_c5_gpu.py is the same loop at 8 x 13 000, message by message.)"""
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R, P, D, CH, K = 8, 125_000, 4096, 1000, 1000
THR, GAP, NB = 0.5, 20, 10
PER_PLACE = 8
OWN = P // PER_PLACE                      # places on a robot's own route
N_PLACES = R * OWN
JUMP = 0.03                               # share of keyframes taken somewhere else (the loop closures)


def _params(r):
    return {"robot_id": r, "max_nb_robots": R, "frontend.sensor_type": "stereo", "frontend.similarity_threshold": THR,
            "frontend.nb_best_matches": NB, "frontend.intra_loop_min_inbetween_keyframes": GAP,
            "frontend.enable_sparsification": True, "evaluation.enable_sparsification_comparison": False}


def _descriptors():
    """Robot r walks its own route, places r OWN .. (r + 1) OWN - 1, 8 keyframes per place, and takes 3 % of its keyframes at a
    random place of ANY route; a visit is the place's unit vector plus noise (same place: cosine ~0.9, different places:
    ~N(0, 1/64)).  Seeded device generators: these are structured test data, not BASELINE.md's random banks (which contain no
    loop closure at all)."""
    import torch
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(2025)
    centres = torch.randn((N_PLACES, D), generator=g, device=dev)
    centres /= centres.norm(dim=1, keepdim=True)
    rng = np.random.default_rng(2025)
    desc = []
    for r in range(R):
        walk = r * OWN + np.arange(P) // PER_PLACE
        walk = np.where(rng.random(P) < JUMP, rng.integers(0, N_PLACES, size=P), walk)
        d = torch.empty((P, D), device=dev)
        for a in range(0, P, 25_000):
            x = centres[torch.from_numpy(walk[a:a + 25_000]).to(dev)] + 0.005 * torch.randn((min(25_000, P - a), D), generator=g, device=dev)
            d[a:a + 25_000] = x / x.norm(dim=1, keepdim=True)
        desc.append(d)
    del centres
    return desc


@pytest.fixture(scope="module")
def c5full():
    import torch
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    from cslam_amd.wire import PackedDescriptorBuffer
    desc = _descriptors()
    host = [d.cpu().numpy() for d in desc]
    lc = [LoopClosureSparseMatching(_params(r)) for r in range(R)]
    bufs = [PackedDescriptorBuffer(r) for r in range(R)]
    rng = np.random.default_rng(7)
    # sampled keyframes per robot: the first chunk (small banks), the last keyframe, and random ones in between
    sampled = {r: sorted({3, CH - 1, P - 1, *rng.integers(CH, P, size=17).tolist()}) for r in range(R)}
    rec = {"intra": {}, "local": {}, "remote": {}}
    n_intra = n_inter = 0
    last = [[-1] * R for _ in range(R)]                                    # last[o][r]: last keyframe of robot r that robot o has received
    t0 = time.perf_counter()
    for s in range(0, P, CH):
        ids = list(range(s, s + CH))
        msgs = []
        for r in range(R):
            intra, inter = lc[r].process_local_keyframes(desc[r][s:s + CH], ids)
            n_intra += sum(k is not None for _, k in intra)
            n_inter += len(inter)
            for j in (j for j in sampled[r] if s <= j < s + CH):
                rec["intra"][(r, j)] = intra[j - s]
                rec["local"][(r, j)] = [tuple(e) for e in inter if e.robot0_keyframe_id == j]
            bufs[r].extend(ids, host[r][s:s + CH])
            msgs.append(list(bufs[r].chunks(s, 10 ** 9)))                  # the step's packed message(s) of robot r
            bufs[r].delete_below(s + CH)
        for o in range(R):                                                 # every receiver drains its queue once per step
            queue = [(c, last[o][r]) for r in range(R) if r != o for c in msgs[r]]
            for (chunk, _), (got, new_last) in zip(queue, lc[o].process_remote_chunks(queue)):
                r = int(chunk.robot_id)
                last[o][r] = new_last
                n_inter += len(got)
                for j in (j for j in sampled[r] if s <= j < s + CH):
                    rec["remote"][(o, r, j)] = [tuple(e) for e in got if e.robot1_keyframe_id == j]
    torch.cuda.synchronize()
    print("C5 front end: 8 x %d keyframes streamed in %.1f s (%d intra, %d inter-robot matches)" % (P, time.perf_counter() - t0, n_intra, n_inter))
    del desc
    torch.cuda.empty_cache()
    return {"lc": lc, "host": host, "sampled": sampled, "rec": rec, "n_intra": n_intra, "n_inter": n_inter}


def _pool_map(fn, items, threads=32):
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(fn, items))


def test_stream_sizes_and_bank_contents(c5full):
    lc = c5full["lc"]
    for r in range(R):
        assert lc[r].local_nnsm.n == P
        assert all(lc[r].other_robots_nnsm[o].n == P for o in range(R) if o != r)
    assert c5full["n_inter"] > 20_000 and c5full["n_intra"] > 1000       # the workload really has loop closures
    sel = lc[0].candidate_selector
    assert len(sel.candidate_edges) > 2 * K
    assert sum(sel.nb_poses.values()) >= 0.99 * R * P          # per robot: 1 + the last keyframe any of robot 0's edges names


def test_intra_decisions_equal_the_oracle_on_sampled_keyframes(c5full):
    from oracle import pyoracle
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    host, rec = c5full["host"], c5full["rec"]

    def want(key):
        r, j = key
        rows, sims, cnt = pyoracle.nns_search(host[r][:max(j, 1)], host[r][j:j + 1], NB, row_limit=np.array([j], dtype=np.int64))
        c = int(cnt[0])
        kfs, s = rows[0, :c].tolist(), sims[0, :c].tolist()              # keyframe id == row in this stream
        return LoopClosureSparseMatching._first_valid(kfs, s, j, GAP, THR) if c else None
    keys = list(rec["intra"])
    for key, w in zip(keys, _pool_map(want, keys)):
        kf_id, got = rec["intra"][key]
        assert kf_id == key[1] and got == w, (key, got, w)
    assert len(keys) >= 8 * 18


def test_inter_robot_best1_equals_the_oracle_both_directions(c5full):
    from oracle import pyoracle
    host, rec = c5full["host"], c5full["rec"]

    def local(key):                                                      # robot r's keyframe j against its copy of every other bank:
        r, j = key                                                       # those hold what was delivered in the earlier steps
        s = j // CH * CH
        out = []
        for o in range(R):
            if o == r or s == 0:
                continue
            rows, sims, cnt = pyoracle.nns_search(host[o][:s], host[r][j:j + 1], 1)
            if cnt[0] > 0 and sims[0, 0] >= THR:
                out.append((r, j, o, int(rows[0, 0]), float(sims[0, 0])))
        return out

    def remote(key):                                                     # robot r's keyframe arriving at robot o, whose local bank
        o, r, j = key                                                    # already holds this step's rows (float64 on receipt)
        s = j // CH * CH
        rows, sims, cnt = pyoracle.nns_search(host[o][:s + CH], host[r][j:j + 1].astype(np.float64), 1)
        return [(o, int(rows[0, 0]), r, j, float(sims[0, 0]))] if cnt[0] > 0 and sims[0, 0] >= THR else []
    checked = matched = 0
    for table, fn in ((rec["local"], local), (rec["remote"], remote)):
        keys = list(table)
        for key, want in zip(keys, _pool_map(fn, keys)):
            got = table[key]
            assert [g[:4] for g in got] == [w[:4] for w in want], (key, got, want)
            assert all(abs(g[4] - w[4]) <= 1e-12 for g, w in zip(got, want))
            checked += 1
            matched += len(want)
    assert checked > 400 and matched > 10                                # the samples do contain matches


def test_candidates_of_the_stream_through_the_million_pose_selection(c5full):
    """What the front end collected, handed to the selection step at the graph's real size."""
    sel = c5full["lc"][0].candidate_selector
    in_range = {r: True for r in range(R)}
    assert sel._fiedler_solver()[0] == "chain_hip" and sum(sel.nb_poses.values()) >= 0.99 * R * P
    before = {sel.edge_key(e) for e in sel.candidate_edges.values()}
    t0 = time.perf_counter()
    first = sel.select_candidates(K, in_range)                           # biased greedy until every robot has a fixed link
    keys1 = {sel.edge_key(e) for e in first}
    assert len(first) == K and len(keys1) == K and keys1 <= before
    sel.candidate_edges_to_fixed(list(first))
    assert all(sel.initial_fixed_edge_exists[r] for r in range(R))
    before = {sel.edge_key(e) for e in sel.candidate_edges.values()}
    second = sel.select_candidates(K, in_range)                          # MAC: Frank-Wolfe over 10^6 poses
    print("C5 front end -> selection: 2 x select_candidates(K = 1000) over %d poses, %d candidates: %.1f s"
          % (sel.total_nb_poses, len(before), time.perf_counter() - t0))
    keys2 = {sel.edge_key(e) for e in second}
    assert 0.99 * R * P <= sel.total_nb_poses <= R * P
    assert len(second) == K and len(keys2) == K and keys2 <= before and not (keys2 & keys1)
    assert not (keys2 & {sel.edge_key(e) for e in sel.candidate_edges.values()})
