"""GPU parity of the sparsifier's device kernels (-m gpu): gradient of lambda_2 and the
Laplacian SpMM, float64, against numpy/scipy (the arithmetic of cslam/mac/mac.py:112-130 and of
the `L @ X` inside networkx's TraceMIN that mac.py:52-58 calls)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_mac_grad_and_csr_spmm_match_numpy():
    import torch
    from cslam_amd import _lib
    from cslam_amd.mac.mac import MAC
    from cslam_amd.mac.utils import Edge
    lib = _lib.load()
    rng = np.random.default_rng(0)
    n, m = 5000, 3000
    fixed = [Edge(k, k + 1, 1.0) for k in range(n - 1)]
    cand = [Edge(int(a), int(b), float(w)) for a, b, w in
            zip(rng.integers(0, n, m), rng.integers(0, n, m), rng.random(m)) if a != b]
    mac = MAC(fixed, cand, n)
    v = rng.standard_normal(n)
    g_ref = mac.grad_from_fiedler(v)
    dv = torch.from_numpy(v).cuda()
    ei = torch.from_numpy(mac.edge_list[:, 0].astype(np.int32)).cuda()
    ej = torch.from_numpy(mac.edge_list[:, 1].astype(np.int32)).cuda()
    w = torch.from_numpy(mac.weights).cuda()
    g = torch.empty(len(cand), dtype=torch.float64, device="cuda")
    _lib.check(lib.cslam_mac_grad_dev(C.c_void_p(dv.data_ptr()), C.c_void_p(ei.data_ptr()), C.c_void_p(ej.data_ptr()),
                                      C.c_void_p(w.data_ptr()), len(cand), C.c_void_p(g.data_ptr()), None))
    assert np.array_equal(g.cpu().numpy(), g_ref)                  # same operation order -> bit-identical

    L = mac.combined_laplacian(rng.random(len(cand))).tocsr()
    L.sort_indices()
    X = rng.standard_normal((n, 4))
    Y_ref = L @ X
    indptr = torch.from_numpy(L.indptr.astype(np.int64)).cuda()
    indices = torch.from_numpy(L.indices.astype(np.int32)).cuda()
    data = torch.from_numpy(L.data).cuda()
    x = torch.from_numpy(np.asfortranarray(X).T.copy()).cuda()     # column-major [n,4] == row-major [4,n]
    y = torch.empty((4, n), dtype=torch.float64, device="cuda")
    _lib.check(lib.cslam_csr_spmm_dev(C.c_void_p(indptr.data_ptr()), C.c_void_p(indices.data_ptr()),
                                      C.c_void_p(data.data_ptr()), n, C.c_void_p(x.data_ptr()), 4,
                                      C.c_void_p(y.data_ptr()), None))
    assert np.max(np.abs(y.cpu().numpy().T - Y_ref)) < 1e-12
    # Laplacian property at any size: rows sum to zero -> L @ 1 == 0
    ones = torch.ones((1, n), dtype=torch.float64, device="cuda")
    y1 = torch.empty((1, n), dtype=torch.float64, device="cuda")
    _lib.check(lib.cslam_csr_spmm_dev(C.c_void_p(indptr.data_ptr()), C.c_void_p(indices.data_ptr()),
                                      C.c_void_p(data.data_ptr()), n, C.c_void_p(ones.data_ptr()), 1,
                                      C.c_void_p(y1.data_ptr()), None))
    assert float(y1.abs().max()) < 1e-9


def _pose_graph(R, P, m, seed):
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    n = R * P
    i = np.concatenate([np.arange(r * P, r * P + P - 1) for r in range(R)])
    fi = np.array([r * P + P - 1 for r in range(R - 1)], dtype=np.int64)
    a, b = rng.integers(0, n, m), rng.integers(0, n, m)
    ok = a != b
    ii = np.concatenate([i, fi, a[ok]]); jj = np.concatenate([i + 1, fi + P, b[ok]])
    ww = np.concatenate([np.ones(len(i)), np.ones(len(fi)), rng.random(ok.sum()) * 0.9 + 0.1])
    rows = np.stack([ii, jj, ii, jj], 1).ravel(); cols = np.stack([ii, jj, jj, ii], 1).ravel()
    data = np.stack([ww, ww, -ww, -ww], 1).ravel()
    return sp.csr_matrix(sp.coo_matrix((data, (rows, cols)), shape=(n, n)))


@pytest.mark.parametrize("R,P,m", [(1, 50, 4), (3, 700, 40), (8, 2600, 900), (2, 4097, 3)])
def test_chain_solver_gpu_matches_host(R, P, m):
    """Segmented-scan kernels + junction solve + back-substitution == the numpy statement
    (chunk boundaries of the 2048-element scan are crossed by segments of every length)."""
    import torch
    from cslam_amd.mac.chain_solver import ChainReducedSolver
    from cslam_amd.mac.chain_solver_gpu import ChainReducedSolverGPU
    L = _pose_graph(R, P, m, R * P + m)
    n = L.shape[0]
    g = int((L.indptr[1:] - L.indptr[:-1]).argmax())
    B = np.random.default_rng(0).standard_normal((n, 4))
    xh = ChainReducedSolver(L, g).solve(B)
    xd = ChainReducedSolverGPU(L, g).solve(torch.from_numpy(B).cuda()).cpu().numpy()
    assert np.max(np.abs(xd - xh)) < 1e-9 * max(1.0, np.max(np.abs(xh)))
    r = L @ xd - B
    r[g] = 0
    assert np.max(np.abs(r)) < 1e-8 * max(1.0, np.max(np.abs(B)))


def test_fiedler_gpu_matches_reference_algorithm():
    from cslam_amd.mac.chain_solver_gpu import fiedler_tracemin_chain_gpu
    from oracle.fiedler_oracle import fiedler_tracemin_lu
    for (R, P, m) in [(3, 100, 30), (8, 400, 600), (8, 2000, 2000)]:
        L = _pose_graph(R, P, m, 7)
        l1, v1 = fiedler_tracemin_lu(L)
        l2, v2 = fiedler_tracemin_chain_gpu(L)
        assert abs(l1 - l2) < 1e-9 * abs(l1) + 1e-13
        assert min(np.max(np.abs(v1 - v2)), np.max(np.abs(v1 + v2))) < 1e-6
        assert np.linalg.norm(L @ v2 - l2 * v2, 1) / abs(L).sum(axis=1).max() < 1e-8


@pytest.mark.parametrize("tag,R,K", [("mac_R3_P100_C100_K10", 3, 10), ("mac_R8_P400_C600_K60", 8, 60)])
def test_selection_with_gpu_solver_equals_reference(tag, R, K):
    from helpers import GOLDEN
    from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot
    g7 = np.load(GOLDEN + "/mac_g7.npz")
    ed = lambda arr: [EdgeInterRobot(int(a), int(b), int(c), int(d), float(w)) for a, b, c, d, w in arr]
    params = {"frontend.enable_sparsification": True, "evaluation.enable_sparsification_comparison": False,
              "frontend.mac_fiedler_solver": "chain_gpu"}
    ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=R, extra_params=params)
    ac.set_graph(ed(g7[tag + "/fixed"]), ed(g7[tag + "/cand"]))
    sel = ac.select_candidates(K, {r: True for r in range(R)}, greedy_initialization=True)
    got = np.array([tuple(e) for e in sel], dtype=np.float64).reshape(-1, 5)
    assert np.array_equal(got, g7[tag + "/selected"])


@pytest.mark.parametrize("m,bs", [(64, 64), (700, 64), (1000, 512), (5000, 2048), (4097, 2048)])
def test_dense_cholesky_solve4_matches_library(m, bs):
    """cslam_chol_solve4_dev (the junction solve of every TraceMIN iteration) against torch.cholesky_solve, for both the
    row-major factor of `blocked_cholesky_` and the column-major one library factorisations return; ragged last block."""
    import torch
    from cslam_amd.mac.chain_solver_gpu import BlockedCholeskySolve, blocked_cholesky_
    g = torch.Generator(device="cuda").manual_seed(m)
    B = torch.randn((m, 96), generator=g, device="cuda", dtype=torch.float64)
    A = B @ B.T + torch.eye(m, device="cuda", dtype=torch.float64) * 50.0
    rhs = torch.randn((m, 4), generator=g, device="cuda", dtype=torch.float64)
    L_lib = torch.linalg.cholesky(A)
    ref = torch.cholesky_solve(rhs, L_lib)
    L_row = blocked_cholesky_(A.clone(), bs=256)                     # upper triangle keeps stale values: must never be read
    L_col = torch.tril(L_lib).T.contiguous().T                       # element (r, c) at c * m + r
    assert L_col.stride(0) == 1 and L_row.stride(1) == 1
    for L in (L_row, L_col):
        x = BlockedCholeskySolve(L, bs).solve(rhs)
        assert float((x - ref).abs().max() / ref.abs().max()) < 1e-11
        assert float((A @ x - rhs).abs().max()) < 1e-9 * float(A.abs().max())
    x2 = BlockedCholeskySolve(L_row, bs).solve(rhs)                   # fixed summation order: bit-identical repeats
    assert torch.equal(x2, BlockedCholeskySolve(L_row, bs).solve(rhs))


def test_chain_gpu_fiedler_pair_is_reproducible_bit_for_bit():
    """Two runs of the HIP Fiedler solver on the same Laplacian give the same bits: the junction matrix is assembled in a
    fixed order (no atomic scatter), every reduction kernel sums in a fixed order, the dense solve has no atomics."""
    from cslam_amd.mac.chain_solver_gpu import fiedler_tracemin_chain_gpu
    L = _pose_graph(4, 6000, 900, 11)
    l1, v1 = fiedler_tracemin_chain_gpu(L)
    l2, v2 = fiedler_tracemin_chain_gpu(L)
    assert l1 == l2 and np.array_equal(v1, v2)


@pytest.mark.parametrize("R,P,m", [(1, 40, 3), (3, 100, 30), (8, 400, 600), (8, 2000, 2000), (2, 4097, 3), (4, 6000, 5200)])
def test_one_call_c_abi_fiedler_matches_reference_algorithm(R, P, m):
    """`cslam_fiedler` (host structure, dense junction factor through rocBLAS / rocSOLVER, TraceMIN loop and its 4 x 4 algebra
    all behind ONE C call) against the sparse-LU restatement of the reference's networkx call (mac.py:35-59) and against
    the torch-driven device solver.  (4, 6000, 5200): > 4096 junctions, the blocked factorisation with 2048 blocks."""
    from cslam_amd.mac.chain_solver_gpu import fiedler_tracemin_chain_gpu, fiedler_tracemin_hip
    from oracle.fiedler_oracle import fiedler_tracemin_lu
    L = _pose_graph(R, P, m, 7)
    l1, v1 = fiedler_tracemin_lu(L)
    st = {}
    l2, v2 = fiedler_tracemin_hip(L, stats=st)
    assert abs(l1 - l2) < 1e-9 * abs(l1) + 1e-13
    assert min(np.max(np.abs(v1 - v2)), np.max(np.abs(v1 + v2))) < 1e-6
    assert np.linalg.norm(L @ v2 - l2 * v2, 1) / abs(L).sum(axis=1).max() < 1e-8
    if L.shape[0] > 4:
        l3, v3 = fiedler_tracemin_chain_gpu(L)
        assert abs(l3 - l2) < 1e-10 * abs(l3) + 1e-13
    # deterministic: no atomics with more than two terms, fixed reduction orders
    l4, v4 = fiedler_tracemin_hip(L)
    assert l4 == l2 and np.array_equal(v4, v2)
    # the caller's start block is honoured: numpy's own block gives the same bits as the built-in generator
    x0 = np.random.RandomState(7).normal(size=(4, L.shape[0])).T
    l5, v5 = fiedler_tracemin_hip(L, x0=x0)
    assert l5 == l2 and np.array_equal(v5, v2)
    assert st['iters'] >= 1


def test_one_call_c_abi_fiedler_argument_errors_and_disconnected_graph():
    import scipy.sparse as sp
    from cslam_amd import _lib
    from cslam_amd._lib import CslamHipError
    from cslam_amd.mac.chain_solver_gpu import fiedler_tracemin_hip
    lib = _lib.load()
    with pytest.raises(CslamHipError, match="n must be"):
        fiedler_tracemin_hip(sp.csr_matrix(np.array([[1.0, -1.0], [-1.0, 1.0]])))
    # two components: the grounded junction Laplacian is singular -> reported, not a wrong answer
    n = 60
    i = np.array([k for k in range(n - 1) if k != 29])
    W = sp.coo_matrix((np.ones(len(i)), (i, i + 1)), shape=(n, n))
    W = W + W.T
    L = sp.csr_matrix(sp.diags(np.asarray(W.sum(axis=1)).ravel()) - W)
    with pytest.raises(CslamHipError, match="not connected"):
        fiedler_tracemin_hip(L, max_iters=50)
    # unsorted column indices are refused
    indptr = np.array([0, 2, 4, 6, 8, 10, 12], dtype=np.int64)
    indices = np.array([1, 0] * 6, dtype=np.int32)
    data = np.ones(12)
    lam, v = C.c_double(), np.empty(6)
    rc = lib.cslam_fiedler(6, indptr.ctypes.data_as(C.c_void_p), indices.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p),
                           None, 7, 1e-8, 0, C.byref(lam), v.ctypes.data_as(C.c_void_p), None, None)
    assert rc == -1 and b"sorted" in lib.cslam_last_error()
    assert lib.cslam_fiedler_release() == 0


@pytest.mark.parametrize("tag,R,K", [("mac_R3_P100_C100_K10", 3, 10), ("mac_R8_P400_C600_K60", 8, 60)])
def test_selection_with_one_call_solver_equals_reference(tag, R, K):
    from helpers import GOLDEN
    from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot
    g7 = np.load(GOLDEN + "/mac_g7.npz")
    ed = lambda arr: [EdgeInterRobot(int(a), int(b), int(c), int(d), float(w)) for a, b, c, d, w in arr]
    params = {"frontend.enable_sparsification": True, "evaluation.enable_sparsification_comparison": False,
              "frontend.mac_fiedler_solver": "chain_hip"}
    ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=R, extra_params=params)
    ac.set_graph(ed(g7[tag + "/fixed"]), ed(g7[tag + "/cand"]))
    sel = ac.select_candidates(K, {r: True for r in range(R)}, greedy_initialization=True)
    got = np.array([tuple(e) for e in sel], dtype=np.float64).reshape(-1, 5)
    assert np.array_equal(got, g7[tag + "/selected"])


@pytest.mark.parametrize("R,P,C_,K", [(3, 100, 100, 10), (4, 300, 500, 40), (8, 400, 600, 60)])
def test_native_frank_wolfe_loop_equals_the_python_loop(R, P, C_, K, monkeypatch):
    """`cslam_mac_fw_subset` (Laplacian updates, Fiedler pairs, gradient, top-k, rounding in native host code around
    cslam_fiedler) against MAC.fw_subset's Python loop (mac.py:191-233 restated) on the same solver: same selection, same
    unrounded iterate and dual bound to round-off."""
    from cslam_amd.mac.mac import MAC
    from cslam_amd.mac.utils import Edge
    rng = np.random.default_rng(R * 1000 + C_)
    n = R * P
    fixed = [Edge(r * P + t, r * P + t + 1, 1.0) for r in range(R) for t in range(P - 1)]
    fixed += [Edge(r * P + int(rng.integers(0, P)), (r + 1) * P + int(rng.integers(0, P)), float(rng.uniform(0.5, 1.0))) for r in range(R - 1)]
    cand = []
    while len(cand) < C_:
        a, b = (int(x) for x in rng.integers(0, n, 2))
        if a // P != b // P:
            cand.append(Edge(min(a, b), max(a, b), float(rng.uniform(0.1, 1.0))))
    mac = MAC(fixed, cand, n, fiedler_solver="chain_hip")
    w0 = np.zeros(len(cand))
    w0[np.argsort([-e.weight for e in cand])[:K]] = 1.0
    sel_c, w_c, u_c = mac.fw_subset(w0, K, max_iters=5)
    monkeypatch.setenv("CSLAM_MAC_FW", "python")
    sel_p, w_p, u_p = mac.fw_subset(w0, K, max_iters=5)
    assert sel_c.sum() == K and np.array_equal(sel_c, sel_p)
    assert np.max(np.abs(w_c - w_p)) < 1e-9 and abs(u_c - u_p) < 1e-9 * max(1.0, abs(u_p))
    # the duality-gap exit: a tolerance no iterate can miss stops after the first Fiedler pair, w untouched
    monkeypatch.delenv("CSLAM_MAC_FW")
    sel1, w1, _ = mac.fw_subset(w0, K, max_iters=5, duality_gap_tol=1e30)
    assert np.array_equal(w1, w0) and sel1.sum() == K


def test_disconnected_start_point_is_retried_like_the_reference_not_raised():
    """Two clusters of robots that only low-weight candidates bridge: the greedy start point leaves the graph disconnected,
    networkx raises there and acm.py:436-466 re-draws the start with one more random pick per trial.  The native solver
    reports that condition as CSLAM_E_GRAPH (CslamGraphError) and takes the same retry path -- a failing kernel would be
    CslamHipError and is never retried away."""
    from cslam_amd._lib import CslamGraphError, CslamHipError
    from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot
    from cslam_amd.mac.chain_solver_gpu import fiedler_tracemin_hip
    import scipy.sparse as sp
    assert issubclass(CslamGraphError, CslamHipError)
    n = 40
    i = np.array([k for k in range(n - 1) if k != 19])
    W = sp.coo_matrix((np.ones(len(i)), (i, i + 1)), shape=(n, n))
    W = W + W.T
    with pytest.raises(CslamGraphError, match="not connected"):
        fiedler_tracemin_hip(sp.csr_matrix(sp.diags(np.asarray(W.sum(axis=1)).ravel()) - W))
    rng = np.random.default_rng(3)
    R, P, K = 4, 40, 6
    fixed = [EdgeInterRobot(0, int(rng.integers(0, P)), 1, int(rng.integers(0, P)), 1.0),
             EdgeInterRobot(2, int(rng.integers(0, P)), 3, int(rng.integers(0, P)), 1.0)]
    cand = []
    for _ in range(30):                                          # heavy candidates inside the two clusters
        a, b = ((0, 1) if rng.random() < 0.5 else (2, 3))
        cand.append(EdgeInterRobot(a, int(rng.integers(0, P)), b, int(rng.integers(0, P)), float(rng.uniform(0.8, 1.0))))
    for _ in range(30):                                          # light candidates across them
        a, b = int(rng.integers(0, 2)), int(rng.integers(2, 4))
        cand.append(EdgeInterRobot(a, int(rng.integers(0, P)), b, int(rng.integers(0, P)), float(rng.uniform(0.05, 0.2))))
    np.random.seed(5)
    params = {"frontend.enable_sparsification": True, "evaluation.enable_sparsification_comparison": False,
              "frontend.mac_fiedler_solver": "chain_hip"}
    ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=R, extra_params=params)
    ac.set_graph(fixed, cand)
    for r in range(R):
        ac.nb_poses[r] = P
    sel = ac.select_candidates(K, {r: True for r in range(R)}, greedy_initialization=True)
    assert len(sel) == K and len({ac.edge_key(e) for e in sel}) == K


@pytest.mark.parametrize("tag,R,K", [("mac_R1_P100_C50_K10", 1, 10), ("mac_R3_P100_C100_K10", 3, 10),
                                     ("mac_R5_P100_C200_K100", 5, 100), ("mac_R8_P400_C600_K60", 8, 60)])
def test_default_solver_on_a_gpu_host_is_the_hip_solver_and_selects_the_reference_edges(tag, R, K):
    """The reference's normal operating regime (a few hundred to a few thousand poses, acm.py:468-543 with default
    parameters): with a GPU visible 'auto' runs `cslam_fiedler` / `cslam_mac_fw_subset`, and the selection is the one the
    reference recorded (G7), edge for edge and in the same order."""
    from helpers import GOLDEN
    from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot
    from cslam_amd.mac import mac as mac_mod
    g7 = np.load(GOLDEN + "/mac_g7.npz")
    ed = lambda arr: [EdgeInterRobot(int(a), int(b), int(c), int(d), float(w)) for a, b, c, d, w in arr]
    ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=R)            # no solver parameter: 'auto'
    assert ac._fiedler_solver() == ("chain_hip", True)
    used = []
    real = mac_mod.MAC._fw_subset_hip

    def spy(self, *a, **kw):
        used.append(self.fiedler_solver)
        return real(self, *a, **kw)
    mac_mod.MAC._fw_subset_hip = spy
    try:
        ac.set_graph(ed(g7[tag + "/fixed"]), ed(g7[tag + "/cand"]))
        sel = ac.select_candidates(K, {r: True for r in range(R)}, greedy_initialization=True)
    finally:
        mac_mod.MAC._fw_subset_hip = real
    got = np.array([tuple(e) for e in sel], dtype=np.float64).reshape(-1, 5)
    assert np.array_equal(got, g7[tag + "/selected"])
    if R > 1:                                        # a single robot never reaches MAC (no fixed inter-robot link)
        assert used and all(u == "chain_hip" for u in used)


def test_auto_falls_back_to_the_torch_driven_solver_without_rocsolver_and_explicit_chain_hip_does_not(monkeypatch):
    """CSLAM_E_UNSUPPORTED (-5: librocblas / librocsolver not found by dlopen) under 'auto' -> 'chain_gpu', same selection;
    an explicitly requested 'chain_hip' keeps raising.  The junction limit has its own code (CSLAM_E_LIMIT, -7)."""
    from helpers import GOLDEN
    from cslam_amd import _lib
    from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot
    from cslam_amd.mac import mac as mac_mod
    from cslam_amd.mac import chain_solver_gpu
    assert issubclass(_lib.CslamUnsupportedError, _lib.CslamHipError) and issubclass(_lib.CslamLimitError, _lib.CslamHipError)
    assert _lib._ERROR_CLASSES[-5] is _lib.CslamUnsupportedError and _lib._ERROR_CLASSES[-7] is _lib.CslamLimitError

    def no_lib(*a, **kw):
        raise _lib.CslamUnsupportedError("libcslam_hip error -5: rocBLAS / rocSOLVER not found")
    monkeypatch.setattr(mac_mod.MAC, "_fw_subset_hip", no_lib)
    monkeypatch.setattr(chain_solver_gpu, "fiedler_tracemin_hip", no_lib)
    g7 = np.load(GOLDEN + "/mac_g7.npz")
    tag, R, K = "mac_R3_P100_C100_K10", 3, 10
    ed = lambda arr: [EdgeInterRobot(int(a), int(b), int(c), int(d), float(w)) for a, b, c, d, w in arr]
    ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=R)
    ac.set_graph(ed(g7[tag + "/fixed"]), ed(g7[tag + "/cand"]))
    sel = ac.select_candidates(K, {r: True for r in range(R)}, greedy_initialization=True)
    assert np.array_equal(np.array([tuple(e) for e in sel], dtype=np.float64).reshape(-1, 5), g7[tag + "/selected"])
    params = {"frontend.enable_sparsification": True, "evaluation.enable_sparsification_comparison": False,
              "frontend.mac_fiedler_solver": "chain_hip"}
    ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=R, extra_params=params)
    ac.set_graph(ed(g7[tag + "/fixed"]), ed(g7[tag + "/cand"]))
    with pytest.raises(_lib.CslamUnsupportedError):
        ac.select_candidates(K, {r: True for r in range(R)}, greedy_initialization=True)


def test_more_junctions_than_the_dense_factor_takes_falls_back_and_still_selects_the_reference_edges(monkeypatch):
    """CSLAM_E_LIMIT (-7): `cslam_fiedler` / `cslam_mac_fw_subset` refuse a graph with more junctions than the dense factor
    takes (64 000 by default: 33 GB); the Python layer then runs the Frank-Wolfe loop itself with the torch-driven solver, whose
    junction system goes through a host sparse LU.  Exercised here with the limit lowered to 50 junctions on a golden graph
    (G7, 8 robots x 400 poses, 600 candidates): the selection must still be the reference's, edge for edge."""
    import ctypes as C
    from helpers import GOLDEN
    from cslam_amd import _lib
    from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot
    from cslam_amd.mac import mac as mac_mod
    from cslam_amd.mac import chain_solver_gpu
    monkeypatch.setenv("CSLAM_FIEDLER_MAX_JUNCTIONS", "50")
    g7 = np.load(GOLDEN + "/mac_g7.npz")
    tag, R, K = "mac_R8_P400_C600_K60", 8, 60
    ed = lambda arr: [EdgeInterRobot(int(a), int(b), int(c), int(d), float(w)) for a, b, c, d, w in arr]
    fell_back = []
    real = chain_solver_gpu.fiedler_tracemin_chain_gpu

    def spy(*a, **kw):
        fell_back.append(1)
        return real(*a, **kw)
    monkeypatch.setattr(chain_solver_gpu, "fiedler_tracemin_chain_gpu", spy)
    ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=R)            # 'auto' -> chain_hip
    ac.set_graph(ed(g7[tag + "/fixed"]), ed(g7[tag + "/cand"]))
    sel = ac.select_candidates(K, {r: True for r in range(R)}, greedy_initialization=True)
    assert np.array_equal(np.array([tuple(e) for e in sel], dtype=np.float64).reshape(-1, 5), g7[tag + "/selected"])
    assert len(fell_back) >= 2, "the junction limit was not hit: the fallback did not run"
    # and the C entry point reports the limit with its own code, not as an invalid argument
    from cslam_amd.mac.mac import MAC
    from cslam_amd.mac.utils import Edge
    n = 2000
    fixed = [Edge(t, t + 1, 1.0) for t in range(n - 1)] + [Edge(7 * t, n - 1 - 5 * t, 0.5) for t in range(1, 150)]
    mac = MAC(fixed, [Edge(3, 1500, 0.7)], n, fiedler_solver="chain_hip")
    with pytest.raises(_lib.CslamLimitError, match="junctions"):
        chain_solver_gpu.fiedler_tracemin_hip(mac.combined_laplacian(np.zeros(1)))
