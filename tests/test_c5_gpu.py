"""BASELINE config 5 at a size that means something (-m gpu): 8 robots x 13 000 keyframes = 104 000 4096-D
descriptors streamed through the batched `LoopClosureSparseMatching` API in the reference's causal order
(gdlcd.py:148-174 per robot; packed wire chunks to the 7 peers, gdlcd.py:198-227 / 407-422), with oracle spot checks
of the intra top-k decision and of both inter-robot best-1 directions on sampled keyframes; then
`select_candidates(K = 1000)` over > 10^5 poses with the chain-reduced HIP Fiedler solver
(acm.py:468-543, mac.py:191-233): selection size, no re-selection, and lambda_2 against the reference's algorithm
(TraceMIN + SuperLU, oracle/fiedler_oracle.py) on the very same Laplacians."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R, P, D, CH, K = 8, 13000, 4096, 500, 1000
THR, GAP, NB = 0.5, 20, 10
N_PLACES, STRIDE = 8000, 1000


def _params(r):
    return {"robot_id": r, "max_nb_robots": R, "frontend.sensor_type": "stereo", "frontend.similarity_threshold": THR,
            "frontend.nb_best_matches": NB, "frontend.intra_loop_min_inbetween_keyframes": GAP,
            "frontend.enable_sparsification": True, "evaluation.enable_sparsification_comparison": False,
            "frontend.mac_fiedler_solver": "chain_gpu"}


def _descriptors():
    """Robot r walks places 1000 r, 1000 r + 1, ... (8 keyframes per place, so neighbouring robots share 625 places at
    different times) and visits a random place on 3 % of its keyframes; a visit is the place's unit vector plus noise
    (same place: cosine ~0.9, different places: ~N(0, 1/64))."""
    import torch
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(55)
    centres = torch.randn((N_PLACES, D), generator=g, device=dev)
    centres /= centres.norm(dim=1, keepdim=True)
    rng = np.random.default_rng(55)
    desc = []
    for r in range(R):
        walk = (STRIDE * r + np.arange(P) // 8) % N_PLACES
        walk = np.where(rng.random(P) < 0.03, rng.integers(0, N_PLACES, size=P), walk)
        d = centres[torch.from_numpy(walk).to(dev)] + 0.005 * torch.randn((P, D), generator=g, device=dev)
        desc.append((d / d.norm(dim=1, keepdim=True)).contiguous())
    return desc


@pytest.fixture(scope="module")
def c5():
    """Runs the stream once; the tests below check different things on the recorded results."""
    import torch
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    from cslam_amd.wire import PackedDescriptorBuffer
    desc = _descriptors()
    host = [d.cpu().numpy() for d in desc]
    lc = [LoopClosureSparseMatching(_params(r)) for r in range(R)]
    bufs = [PackedDescriptorBuffer(r) for r in range(R)]
    rng = np.random.default_rng(7)
    # sampled keyframes per robot: the first chunk (small banks), the last, and random ones in between
    sampled = {r: sorted({3, CH - 1, P - 1, *rng.integers(CH, P, size=5).tolist()}) for r in range(R)}
    rec = {"intra": {}, "local": {}, "remote": {}}
    n_intra = n_inter = 0
    for s in range(0, P, CH):
        ids = list(range(s, s + CH))
        for r in range(R):
            intra, inter = lc[r].process_local_keyframes(desc[r][s:s + CH], ids)
            n_intra += sum(k is not None for _, k in intra)
            n_inter += len(inter)
            want = [j for j in sampled[r] if s <= j < s + CH]
            for j in want:
                rec["intra"][(r, j)] = intra[j - s]
                rec["local"][(r, j)] = [tuple(e) for e in inter if e.robot0_keyframe_id == j]
            bufs[r].extend(ids, host[r][s:s + CH])
            for chunk in bufs[r].chunks(s, 10 ** 9):                       # one packed message per chunk of keyframes
                for o in range(R):
                    if o == r:
                        continue
                    got, _ = lc[o].process_remote_chunk(chunk, s - 1)
                    n_inter += len(got)
                    for j in want:
                        rec["remote"][(o, r, j)] = [tuple(e) for e in got if e.robot1_keyframe_id == j]
            bufs[r].delete_below(s + CH)
    torch.cuda.synchronize()
    return {"lc": lc, "host": host, "sampled": sampled, "rec": rec, "n_intra": n_intra, "n_inter": n_inter}


def _visible(o, r, s):
    """keyframes of robot o that exist anywhere when robot r takes its turn in the step starting at s"""
    return s + CH if o < r else s


def test_stream_sizes_and_bank_contents(c5):
    lc = c5["lc"]
    for r in range(R):
        assert lc[r].local_nnsm.n == P
        assert all(lc[r].other_robots_nnsm[o].n == P for o in range(R) if o != r)
    assert c5["n_inter"] > 20000 and c5["n_intra"] > 1000          # the workload really has loop closures
    sel = lc[0].candidate_selector
    assert len(sel.candidate_edges) > 2 * K
    assert sum(sel.nb_poses.values()) >= 100_000


def test_intra_decisions_equal_the_oracle_on_sampled_keyframes(c5):
    from oracle import pyoracle
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    host, rec = c5["host"], c5["rec"]
    for (r, j), (kf_id, got) in rec["intra"].items():
        assert kf_id == j
        rows, sims, cnt = pyoracle.nns_search(host[r][:max(j, 1)], host[r][j:j + 1], NB,
                                              row_limit=np.array([j], dtype=np.int64))
        c = int(cnt[0])
        kfs, s = rows[0, :c].tolist(), sims[0, :c].tolist()          # keyframe id == row in this stream
        want = LoopClosureSparseMatching._first_valid(kfs, s, j, GAP, THR) if c else None
        assert got == want, (r, j, got, want, kfs, s)


def test_inter_robot_best1_equals_the_oracle_both_directions(c5):
    from oracle import pyoracle
    host, rec = c5["host"], c5["rec"]
    checked = matched = 0
    for (r, j), got in rec["local"].items():                         # local keyframe vs every other robot's bank
        s = j // CH * CH
        want = []
        for o in range(R):
            n_o = _visible(o, r, s)
            if o == r or n_o == 0:
                continue
            rows, sims, cnt = pyoracle.nns_search(host[o][:n_o], host[r][j:j + 1], 1)
            if cnt[0] > 0 and sims[0, 0] >= THR:
                want.append((r, j, o, int(rows[0, 0]), float(sims[0, 0])))
        assert [g[:4] for g in got] == [w[:4] for w in want], (r, j, got, want)
        assert all(abs(g[4] - w[4]) <= 1e-12 for g, w in zip(got, want))
        checked += 1
        matched += len(want)
    for (o, r, j), got in rec["remote"].items():                     # robot r's keyframe arriving at robot o
        s = j // CH * CH
        n_o = _visible(o, r, s)
        want = []
        if n_o > 0:
            rows, sims, cnt = pyoracle.nns_search(host[o][:n_o], host[r][j:j + 1].astype(np.float64), 1)
            if cnt[0] > 0 and sims[0, 0] >= THR:
                want.append((o, int(rows[0, 0]), r, j, float(sims[0, 0])))
        assert [g[:4] for g in got] == [w[:4] for w in want], (o, r, j, got, want)
        assert all(abs(g[4] - w[4]) <= 1e-12 for g, w in zip(got, want))
        checked += 1
        matched += len(want)
    assert checked > 400 and matched > 20                            # the samples do contain matches


def test_select_candidates_1000_of_100k_poses_chain_gpu(c5, monkeypatch):
    from cslam_amd.mac import mac as mac_mod
    from cslam_amd.mac.chain_solver_gpu import fiedler_tracemin_hip
    from oracle.fiedler_oracle import fiedler_tracemin_lu
    sel = c5["lc"][0].candidate_selector
    in_range = {r: True for r in range(R)}
    before = {sel.edge_key(e) for e in sel.candidate_edges.values()}
    first = sel.select_candidates(K, in_range)                       # biased greedy until every robot has a fixed link
    assert len(first) == K and len({sel.edge_key(e) for e in first}) == K
    assert {sel.edge_key(e) for e in first} <= before
    sel.candidate_edges_to_fixed(list(first))
    assert all(sel.initial_fixed_edge_exists[r] for r in range(R))

    seen = []
    real = mac_mod.MAC.evaluate_fiedler_pair

    def spy(self, w, *a, **kw):
        f, v = real(self, w, *a, **kw)
        seen.append((self, np.array(w, copy=True), float(f), np.array(v, copy=True)))
        return f, v
    monkeypatch.setattr(mac_mod.MAC, "evaluate_fiedler_pair", spy)
    before = {sel.edge_key(e) for e in sel.candidate_edges.values()}
    second = sel.select_candidates(K, in_range)                      # MAC: Frank-Wolfe over > 10^5 poses
    monkeypatch.undo()
    assert sel._fiedler_solver() == ("chain_gpu", False) and sel.total_nb_poses >= 100_000
    assert len(seen) >= 2, "the Frank-Wolfe loop did not run (fell back to greedy?)"
    assert all(m.fiedler_solver == "chain_gpu" and m.num_poses == sel.total_nb_poses for m, _, _, _ in seen)
    keys2 = {sel.edge_key(e) for e in second}
    assert len(second) == K and len(keys2) == K and keys2 <= before
    assert not (keys2 & {sel.edge_key(e) for e in first}), "an edge was selected twice"
    assert not (keys2 & {sel.edge_key(e) for e in sel.candidate_edges.values()})     # removed from the candidates
    # lambda_2 of the HIP solver vs the reference's algorithm on the same Laplacian: the first iterate (K one-hot
    # weights) and the second (fractional weights, up to 2K candidate edges in L)
    for m, w, f, v in seen[:2]:
        L = m.combined_laplacian(w)
        f_ref, v_ref = fiedler_tracemin_lu(L, tol=1e-8, seed=np.random.RandomState(7))
        assert f > 0 and abs(f - f_ref) <= 1e-9 * abs(f_ref), (f, f_ref)
        # the Fiedler vector up to sign (gradient uses (v_i - v_j)^2): both unit norm
        assert min(np.abs(v - v_ref).max(), np.abs(v + v_ref).max()) <= 1e-6
        # and the C ABI's one-call solver (what a host without Python runs in place of mac.py:35-59) on the same Laplacian
        f_c, v_c = fiedler_tracemin_hip(L)
        assert abs(f_c - f_ref) <= 1e-9 * abs(f_ref), (f_c, f_ref)
        assert min(np.abs(v_c - v_ref).max(), np.abs(v_c + v_ref).max()) <= 1e-6
    # the objective never decreases below the first iterate's by more than round-off at the final rounding: the
    # selection is at least as connected as the greedy start
    sel.candidate_edges_to_fixed(list(second))
    third = sel.select_candidates(50, in_range)
    assert not ({sel.edge_key(e) for e in third} & (keys2 | {sel.edge_key(e) for e in first}))
