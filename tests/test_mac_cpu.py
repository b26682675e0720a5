"""CPU suite: candidate sparsifier host logic (cslam_amd.algebraic_connectivity_maximization,
cslam_amd.mac) against golden vectors recorded from the REFERENCE classes
(tests/golden/mac_g7.npz, made by oracle/gen_golden_mac.py), plus the reference's own
property tests (tests/test_algebraic_connectivity.py) re-expressed for this package.
"""
import random

import numpy as np
import pytest

from helpers import GOLDEN
from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot
from cslam_amd.mac.mac import MAC
from cslam_amd.mac.utils import Edge, weight_graph_lap_from_edge_list


@pytest.fixture(scope="module")
def g7():
    return np.load(GOLDEN + "/mac_g7.npz")


def _edges(arr):
    return [EdgeInterRobot(int(a), int(b), int(c), int(d), float(w)) for a, b, c, d, w in arr]


@pytest.mark.parametrize("tag,R,K", [("mac_R1_P100_C50_K10", 1, 10), ("mac_R3_P100_C100_K10", 3, 10),
                                     ("mac_R5_P100_C200_K100", 5, 100), ("mac_R8_P400_C600_K60", 8, 60)])
def test_selection_identical_to_reference(g7, tag, R, K):
    trace = []
    orig = MAC.fw_subset

    def traced(self, w_init, k, max_iters=5, duality_gap_tol=1e-8, trace=None):
        return orig(self, w_init, k, max_iters=max_iters, duality_gap_tol=duality_gap_tol, trace=trace_sink)

    trace_sink = trace
    MAC.fw_subset = traced
    try:
        ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=R)
        ac.set_graph(_edges(g7[tag + "/fixed"]), _edges(g7[tag + "/cand"]))
        sel = ac.select_candidates(K, {r: True for r in range(R)}, greedy_initialization=True)
    finally:
        MAC.fw_subset = orig
    got = np.array([tuple(e) for e in sel], dtype=np.float64).reshape(-1, 5)
    assert np.array_equal(got, g7[tag + "/selected"])                       # same edges, same order
    ref_trace = g7[tag + "/trace"]
    assert len(trace) == len(ref_trace)
    if len(trace):
        t = np.array(trace)
        assert np.allclose(t[:, 0], ref_trace[:, 0], rtol=1e-9, atol=1e-13)  # lambda_2 per FW iteration
        assert np.allclose(t[:, 1], ref_trace[:, 1], rtol=1e-7, atol=1e-13)  # ||grad||
    assert np.array_equal(np.array(sorted(ac.candidate_edges.keys()), dtype=np.int64).reshape(-1, 4),
                          g7[tag + "/remaining"])


def test_bookkeeping_script_identical_to_reference(g7):
    R = 3
    ac = AlgebraicConnectivityMaximization(robot_id=1, max_nb_robots=R)
    matches = [tuple(m) for m in g7["acm/matches"]]
    for m in matches:
        ac.add_match(EdgeInterRobot(int(m[0]), int(m[1]), int(m[2]), int(m[3]), float(m[4])))
    for m in matches[:10]:
        ac.add_match(EdgeInterRobot(int(m[2]), int(m[3]), int(m[0]), int(m[1]), float(m[4]) + 0.5))
        ac.add_match(EdgeInterRobot(int(m[0]), int(m[1]), int(m[2]), int(m[3]), float(m[4]) - 0.5))

    def check(step):
        keys = sorted(ac.candidate_edges.keys())
        assert np.array_equal(np.array(keys, dtype=np.int64).reshape(-1, 4), g7[f"acm/s{step}_keys"])
        assert np.array_equal(np.array([ac.candidate_edges[k].weight for k in keys]), g7[f"acm/s{step}_w"])
        assert [ac.nb_poses[r] for r in range(R)] == list(g7[f"acm/s{step}_nb_poses"])
        assert [ac.initial_fixed_edge_exists[r] for r in range(R)] == list(g7[f"acm/s{step}_ife"])
        assert len(ac.fixed_edges) == int(g7[f"acm/s{step}_nfixed"])
        assert np.array_equal(np.array(sorted(ac.already_considered_matches), dtype=np.int64).reshape(-1, 4),
                              g7[f"acm/s{step}_considered"])

    def same(sel, name):
        assert np.array_equal(np.array([tuple(e) for e in sel], dtype=np.float64).reshape(-1, 5), g7[name])

    check(0)
    inr = {0: True, 1: True, 2: True}
    sel0 = ac.select_candidates(5, inr); same(sel0, "acm/sel0"); check(1)
    ac.candidate_edges_to_fixed(list(sel0[:3]))
    ac.remove_candidate_edges(list(sel0[3:]), failed=True); check(2)
    sel1 = ac.select_candidates(4, inr); same(sel1, "acm/sel1"); check(3)
    sel2 = ac.select_candidates(4, {0: True, 1: True, 2: False}); same(sel2, "acm/sel2"); check(4)
    ac.compute_offsets(ac.check_graph_disconnections({0: True, 1: True, 2: False}))
    assert [ac.offsets[r] for r in range(R)] == list(g7["acm/final_offsets"])


# ---- the reference's property tests, re-expressed (tests/test_algebraic_connectivity.py) ----
def build_multi_robot_graph(nb_poses, nb_candidate_edges, max_nb_robots, rnd):
    fixed = [EdgeInterRobot(i, nb_poses - 1, i + 1, nb_poses - 1, 1) for i in range(max_nb_robots - 1)]
    cand = {}
    while len(cand) < nb_candidate_edges:
        r0 = rnd.randrange(max_nb_robots)
        r1 = rnd.choice([r for r in range(max_nb_robots) if r != r0])
        e = EdgeInterRobot(r0, rnd.randrange(nb_poses), r1, rnd.randrange(nb_poses), 1)
        key = (e.robot0_id, e.robot0_keyframe_id, e.robot1_id, e.robot1_keyframe_id) if r0 < r1 else \
              (e.robot1_id, e.robot1_keyframe_id, e.robot0_id, e.robot0_keyframe_id)
        cand[key] = e
    return fixed, list(cand.values())


def build_simple_graph(nb_poses, nb_candidate_edges, rnd):
    cand = {}
    while len(cand) < nb_candidate_edges:
        e = EdgeInterRobot(0, rnd.randrange(nb_poses), 0, rnd.randrange(nb_poses), 1)
        cand[(e.robot0_keyframe_id, e.robot1_keyframe_id)] = e
    return [], list(cand.values())


def test_edge_equality_ignores_weight_and_direction():
    a = EdgeInterRobot(0, 1, 2, 3, 0.5)
    assert a == EdgeInterRobot(0, 1, 2, 3, 0.9) and a == EdgeInterRobot(2, 3, 0, 1, 0.1)
    assert not (a == EdgeInterRobot(0, 1, 2, 4, 0.5))
    assert a in [EdgeInterRobot(2, 3, 0, 1, 7.0)]


def test_selection_sizes_and_no_reselection():
    rnd = random.Random(0)
    np.random.seed(0)
    fixed, cand = build_simple_graph(100, 50, rnd)
    ac = AlgebraicConnectivityMaximization()
    ac.set_graph(fixed, cand)
    before = list(ac.candidate_edges.values())
    sel0 = ac.select_candidates(10, {0: True}, greedy_initialization=False)
    assert len(sel0) == 10 and len(set(sel0)) == 10
    assert all(e in before for e in sel0)
    ac.candidate_edges_to_fixed(sel0)
    assert all(e not in list(ac.candidate_edges.values()) for e in sel0)
    sel1 = ac.select_candidates(10, {0: True}, greedy_initialization=False)
    for e0 in sel0:
        for e1 in sel1:
            assert not (e0.robot0_keyframe_id == e1.robot0_keyframe_id and
                        e0.robot1_keyframe_id == e1.robot1_keyframe_id)
    for i in range(10):
        ac.add_candidate_edge(EdgeInterRobot(0, rnd.randrange(100), 0, rnd.randrange(100), 1.0))
    assert len(ac.select_candidates(12, {0: True}, greedy_initialization=False)) == 12


def test_greedy_initialization_is_topk_weight_sum():
    rnd = random.Random(1)
    fixed, cand = build_simple_graph(100, 50, rnd)
    weights = np.random.default_rng(1).random(50)
    ac = AlgebraicConnectivityMaximization()
    cand = [ac.replace_weight(e, weight=w) for e, w in zip(cand, weights)]
    ac.set_graph(fixed, cand)
    inc = ac.check_graph_disconnections({0: True})
    ac.compute_offsets(inc)
    edges = ac.rekey_edges(ac.candidate_edges.values(), inc)
    w_init = ac.greedy_initialization(10, edges)
    assert abs(np.sum(weights[w_init.astype(bool)]) - np.sum(np.sort(weights)[-10:])) < 1e-12


def test_offsets_rekey_roundtrip_and_exclusion():
    rnd = random.Random(2)
    fixed, cand = build_multi_robot_graph(10, 10, 5, rnd)
    ac = AlgebraicConnectivityMaximization(robot_id=1, max_nb_robots=5)
    ac.set_graph(fixed, cand)
    considered = {i: True for i in range(5)}
    inc = ac.check_graph_disconnections(considered)
    assert all(inc.values())
    ac.compute_offsets(inc)
    for r in range(1, 5):
        assert ac.offsets[r] == ac.offsets[r - 1] + ac.nb_poses[r - 1]
    rek = ac.rekey_edges(ac.candidate_edges.values(), inc)
    back = ac.recover_inter_robot_edges(rek, inc)
    assert back == list(ac.candidate_edges.values())
    for e in rek:
        assert 0 <= e.i < sum(ac.nb_poses.values()) and 0 <= e.j < sum(ac.nb_poses.values())
    considered[2] = False
    inc = ac.check_graph_disconnections(considered)
    assert not inc[2] and inc[0] and inc[1] and inc[3] and inc[4]
    ac.compute_offsets(inc)
    assert ac.offsets[2] == 0 and ac.offsets[3] == ac.offsets[1] + ac.nb_poses[1]
    assert all(e.robot0_id != 2 and e.robot1_id != 2 for e in
               ac.get_included_edges(ac.candidate_edges.values(), inc))


def test_add_match_keeps_max_weight_only_for_ordered_key():
    """The reference quirk (acm.py:559-572): the lookup key is not direction-normalised."""
    ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=2)
    ac.add_match(EdgeInterRobot(0, 1, 1, 2, 0.5))
    ac.add_match(EdgeInterRobot(0, 1, 1, 2, 0.3))
    assert ac.candidate_edges[(0, 1, 1, 2)].weight == 0.5
    ac.add_match(EdgeInterRobot(0, 1, 1, 2, 0.8))
    assert ac.candidate_edges[(0, 1, 1, 2)].weight == 0.8
    ac.add_match(EdgeInterRobot(1, 2, 0, 1, 0.1))      # reversed direction: overwrites (quirk)
    assert ac.candidate_edges[(0, 1, 1, 2)].weight == 0.1


def test_remove_candidates_and_never_reconsider():
    rnd = random.Random(3)
    fixed, cand = build_multi_robot_graph(10, 10, 3, rnd)
    ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=3)
    ac.set_graph(fixed, cand)
    n0 = len(ac.candidate_edges)
    first = list(ac.candidate_edges.values())[0]
    ac.remove_candidate_edges([first])
    assert len(ac.candidate_edges) == n0 - 1
    ac.remove_candidate_edges([EdgeInterRobot(0, 1, 4, 1, 1.0)])
    assert len(ac.candidate_edges) == n0 - 1
    ac.add_candidate_edge(first)                       # already considered -> ignored
    assert len(ac.candidate_edges) == n0 - 1


def test_laplacian_assembly_and_gradient():
    edges = [Edge(0, 1, 1.0), Edge(1, 2, 2.0), Edge(0, 2, 0.5), Edge(1, 2, 0.25)]
    L = weight_graph_lap_from_edge_list(edges, 4).toarray()
    ref = np.zeros((4, 4))
    for e in edges:
        ref[e.i, e.i] += e.weight; ref[e.j, e.j] += e.weight
        ref[e.i, e.j] -= e.weight; ref[e.j, e.i] -= e.weight
    assert np.array_equal(L, ref)
    mac = MAC([Edge(k, k + 1, 1.0) for k in range(9)], [Edge(0, 9, 0.7), Edge(2, 6, 0.4)], 10)
    lam, v = mac.evaluate_fiedler_pair(np.array([1.0, 0.0]))
    Ld = mac.combined_laplacian(np.array([1.0, 0.0])).toarray()
    w = np.linalg.eigvalsh(Ld)
    assert abs(lam - w[1]) < 1e-8 and np.linalg.norm(Ld @ v - lam * v) < 1e-6
    g = mac.grad_from_fiedler(v)
    assert np.allclose(g, [0.7 * (v[0] - v[9]) ** 2, 0.4 * (v[2] - v[6]) ** 2])


# ---- chain-reduced solver (cslam_amd/mac/chain_solver.py): exact elimination of the odometry chains ----
def _random_pose_graph(R, P, m, seed, overlap=False):
    rng = np.random.default_rng(seed)
    edges = [Edge(r * P + k, r * P + k + 1, 1.0) for r in range(R) for k in range(P - 1)]
    edges += [Edge(r * P + P - 1, (r + 1) * P + P - 1, 1.0) for r in range(R - 1)]
    if overlap:                      # the reference chains excluded robots at offset 0: doubled chain edges
        edges += [Edge(k, k + 1, 1.0) for k in range(P // 2)]
    for _ in range(m):
        a, b = rng.integers(0, R * P, 2)
        if a != b:
            edges.append(Edge(int(a), int(b), float(rng.random() * 0.9 + 0.1)))
    return weight_graph_lap_from_edge_list(edges, R * P)


@pytest.mark.parametrize("R,P,m,overlap", [(1, 60, 5, False), (3, 50, 12, False), (4, 300, 80, True), (8, 400, 600, False)])
def test_chain_reduced_solver_is_exact(R, P, m, overlap):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from cslam_amd.mac.chain_solver import ChainReducedSolver, fiedler_tracemin_chain
    from oracle.fiedler_oracle import fiedler_tracemin_lu
    L = _random_pose_graph(R, P, m, R + P, overlap)
    n = L.shape[0]
    g = int((L.indptr[1:] - L.indptr[:-1]).argmax())
    keep = np.array([i for i in range(n) if i != g])
    B = np.random.default_rng(1).standard_normal((n, 4))
    xr = np.zeros((n, 4))
    xr[keep] = spla.splu(sp.csc_matrix(L)[keep][:, keep].tocsc()).solve(B[keep])
    x = ChainReducedSolver(L, g).solve(B)
    assert np.max(np.abs(x - xr)) < 1e-9 * max(1.0, np.max(np.abs(xr)))
    assert np.all(x[g] == 0)
    l1, v1 = fiedler_tracemin_lu(L)
    l2, v2 = fiedler_tracemin_chain(L)
    assert abs(l1 - l2) < 1e-12 * max(1.0, abs(l1)) + 1e-14
    assert min(np.max(np.abs(v1 - v2)), np.max(np.abs(v1 + v2))) < 1e-9


def test_tracemin_restatement_and_product_solver_equal_the_reference_dependency():
    """oracle/fiedler_oracle.py restates networkx's private TraceMIN (the function cslam/mac/mac.py:35-59 calls); the product's
    'tracemin_lu' solver CALLS it.  Same start block, same iterates: lambda_2 and the Fiedler vector agree to round-off."""
    pytest.importorskip("networkx")
    from cslam_amd.mac.fiedler import fiedler_tracemin_lu as product
    from oracle.fiedler_oracle import fiedler_tracemin_lu as restated
    for R, P, m in ((3, 50, 12), (8, 400, 600)):
        L = _random_pose_graph(R, P, m, R + P, False)
        l1, v1 = product(L)
        l2, v2 = restated(L)
        assert abs(l1 - l2) <= 1e-13 * max(1.0, abs(l1))
        assert min(np.max(np.abs(v1 - v2)), np.max(np.abs(v1 + v2))) < 1e-10


def test_chain_solver_handles_missing_chain_edges_and_ragged_ids():
    """Nodes that are not consecutive on any chain (gaps between robots) are junctions."""
    from cslam_amd.mac.chain_solver import ChainReducedSolver
    edges = [Edge(k, k + 1, 2.0) for k in range(0, 19)] + [Edge(k, k + 1, 0.5) for k in range(20, 44)]
    edges += [Edge(5, 30, 1.0), Edge(19, 20 + 7, 0.3), Edge(0, 44, 0.7), Edge(2, 4, 0.9)]
    L = weight_graph_lap_from_edge_list(edges, 45)
    Ld = L.toarray()
    g = 30
    keep = [i for i in range(45) if i != g]
    b = np.random.default_rng(2).standard_normal(45)
    xr = np.zeros(45); xr[keep] = np.linalg.solve(Ld[np.ix_(keep, keep)], b[keep])
    x = ChainReducedSolver(L, g).solve(b)
    assert np.max(np.abs(x - xr)) < 1e-11


@pytest.mark.parametrize("tag,R,K", [("mac_R3_P100_C100_K10", 3, 10), ("mac_R5_P100_C200_K100", 5, 100),
                                     ("mac_R8_P400_C600_K60", 8, 60)])
def test_selection_with_chain_solver_equals_reference(g7, tag, R, K):
    """MAC driven by the chain-reduced (host) solver selects exactly the reference's edges."""
    params = {"frontend.enable_sparsification": True, "evaluation.enable_sparsification_comparison": False,
              "frontend.mac_fiedler_solver": "chain"}
    ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=R, extra_params=params)
    ac.set_graph(_edges(g7[tag + "/fixed"]), _edges(g7[tag + "/cand"]))
    sel = ac.select_candidates(K, {r: True for r in range(R)}, greedy_initialization=True)
    got = np.array([tuple(e) for e in sel], dtype=np.float64).reshape(-1, 5)
    assert np.array_equal(got, g7[tag + "/selected"])


def test_c_abi_start_block_is_numpy_randomstate_normal_bit_for_bit():
    """cslam_fiedler's built-in start block (host code: MT19937 + the legacy polar Box-Muller) equals
    np.random.RandomState(seed).normal(size=(4, n)).T, the block networkx draws for the reference (mac.py:56-58)."""
    import ctypes as C
    from cslam_amd import _lib
    lib = _lib.load()
    for seed, n in ((7, 1), (7, 1237), (7, 50001), (2**32 - 1, 333), (0, 64)):
        x = np.empty((n, 4))
        assert lib.cslam_fiedler_start_block(seed, n, x.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(x, np.random.RandomState(seed).normal(size=(4, n)).T)
    assert lib.cslam_fiedler_start_block(7, 0, None) == -1


def test_tracemin_lu_without_networkx_falls_back_to_the_chain_solver_and_says_so(monkeypatch):
    """A host whose networkx is missing, or has moved the private `_get_fiedler_func`, must not end in run_mac_solver's silent
    retry-then-pseudo-greedy path: the product's 'tracemin_lu' then runs the same TraceMIN on chain_solver.py's inner solves
    (numpy / scipy only) and warns once; the pair equals the one networkx returns."""
    import warnings
    import scipy.sparse as sp
    from networkx.linalg import algebraicconnectivity as nxac
    from cslam_amd.mac import fiedler as fmod
    rng = np.random.default_rng(5)
    n = 60
    rows = list(range(n - 1)) + list(rng.integers(0, n, size=25))
    cols = list(range(1, n)) + list(rng.integers(0, n, size=25))
    w = rng.uniform(0.2, 1.0, size=len(rows))
    A = sp.coo_matrix((w, (rows, cols)), shape=(n, n)).tocsr()
    A = A + A.T
    A.setdiag(0)
    L = sp.csr_matrix(sp.diags(np.asarray(A.sum(axis=1)).ravel()) - A)
    lam_nx, v_nx = fmod.fiedler_tracemin_lu(L)

    def gone(name):
        raise AttributeError("module has no attribute '_get_fiedler_func'")
    monkeypatch.setattr(nxac, "_get_fiedler_func", gone)
    monkeypatch.setattr(fmod, "_warned", [])
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        lam, v = fmod.fiedler_tracemin_lu(L)
        lam2, _ = fmod.fiedler_tracemin_lu(L)
    assert sum("chain-reduced TraceMIN" in str(r.message) for r in rec) == 1
    assert abs(lam - lam_nx) <= 1e-9 * abs(lam_nx) and lam2 == lam
    assert min(np.abs(v - v_nx).max(), np.abs(v + v_nx).max()) <= 1e-6


def test_run_mac_solver_does_not_retry_away_a_missing_dependency(monkeypatch):
    """ImportError / AttributeError out of the solver are programming or packaging errors, not "the Laplacian of this start point
    is singular": they are re-raised, not converted into K retries and the pseudo-greedy selection."""
    from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot
    from cslam_amd.mac import mac as mac_mod
    ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=2)
    fixed = [EdgeInterRobot(0, 3, 1, 3, 1.0)]
    cand = [EdgeInterRobot(0, i, 1, (i * 3) % 7, 0.5 + 0.05 * i) for i in range(6)]
    ac.set_graph(fixed, cand)

    def broken(self, *a, **k):
        raise ImportError("No module named 'networkx'")
    monkeypatch.setattr(mac_mod.MAC, "fw_subset", broken)
    with pytest.raises(ImportError):
        ac.select_candidates(2, {0: True, 1: True})
