"""TEST INFRASTRUCTURE (checker of SURVEY 8 row a23): Fiedler pair of a graph Laplacian by TraceMIN with sparse-LU inner solves.
Only tests/ import this module; the product's solvers are cslam_amd/mac/{fiedler,chain_solver,chain_solver_gpu}.py.

The reference obtains it from a PRIVATE networkx function,
    la.algebraicconnectivity._get_fiedler_func('tracemin_lu')(L, x=None, normalized=False,
                                                   tol=1e-8, seed=np.random.RandomState(7))
(cslam/mac/mac.py:35-59; networkx 2.7/2.8 pinned by the reference, 3.4.2 in the build
container; source not under /root/reference).  This is a restatement of that published
algorithm (Manguoglu, Cox, Saied, Sameh: "TRACEMIN-Fiedler", 2010) with the same start
block, the same stopping rule and the same SuperLU options, so the iterates coincide:
    X0 = RandomState(7).normal(size=(q, n)).T, q = min(4, n-1); project out the constant vector
    loop: X = qr(X).Q ; W = L X ; H = X'W ; (sigma, Y) = eigh(H) ; X = X Y
          stop when ||W Y[:,0] - sigma0 X[:,0]||_1 / ||L||_inf < tol
          W = A^-1 X  with A = L except A[i,i] = inf at the densest column (forces x_i = 0)
          X = (inv(W'X) W')' ; project
Pinned by tests/golden/mac_g7.npz (lambda_2 per Frank-Wolfe iteration of the reference, generated with networkx 3.4.2 by
oracle/gen_golden_mac.py) and, where networkx is importable, against the private function itself (tests/test_mac_cpu.py).
"""
import numpy as np
import scipy as sp
import scipy.linalg
import scipy.sparse
import scipy.sparse.linalg


def _project(X, n):
    for j in range(X.shape[1]):
        X[:, j] -= X[:, j].sum() / n


def fiedler_tracemin_lu(L, tol=1e-8, seed=None):
    """Returns (lambda_2, v_2) of the (connected, unnormalised) Laplacian L."""
    if seed is None:
        seed = np.random.RandomState(7)
    n = L.shape[0]
    q = min(4, n - 1)
    X = np.asarray(seed.normal(size=(q, n))).T

    A = sp.sparse.csc_array(L, dtype=float, copy=True)
    i = (A.indptr[1:] - A.indptr[:-1]).argmax()
    A[i, i] = np.inf
    lu = sp.sparse.linalg.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
                               options={"Equil": True, "SymmetricMode": True})

    Lnorm = abs(L).sum(axis=1).flatten().max()
    _project(X, n)
    W = np.ndarray(X.shape, order="F")
    while True:
        X = np.linalg.qr(X)[0]
        W[:, :] = L @ X
        H = X.T @ W
        sigma, Y = sp.linalg.eigh(H, overwrite_a=True)
        X = X @ Y
        res = sp.linalg.blas.dasum(W @ Y[:, 0] - sigma[0] * X[:, 0]) / Lnorm
        if res < tol:
            break
        for j in range(X.shape[1]):
            W[:, j] = lu.solve(X[:, j])
        X = (sp.linalg.inv(W.T @ X) @ W.T).T
        _project(X, n)
    return sigma[0], np.asarray(X)[:, 0]
