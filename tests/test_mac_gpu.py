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
