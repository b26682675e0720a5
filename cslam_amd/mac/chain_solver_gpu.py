"""Device version of the chain-reduced Fiedler solver (see chain_solver.py for the algorithm).

Per Laplacian: the host builds the chain/junction structure (O(n) numpy), the reduced junction
Laplacian is factorised densely on the GPU (float64 Cholesky), and every TraceMIN iteration runs
    W = L X                      cslam_csr_spmm4_dev      (HIP, one lane per row)
    W = A^-1 X                   cslam_chain_forward_dev  (two segmented scans + junction gather, HIP)
                                 dense triangular solves on the junction system
                                 cslam_chain_backward_dev (closed-form interior potentials, HIP)
plus 4x4 dense algebra.  Vectors are [n][4] float64 and never leave HBM.
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from .. import _lib
from .chain_solver import ChainReducedSolver


def _p(t):
    return C.c_void_p(t.data_ptr())


def blocked_cholesky_(A, bs=2048):
    """In-place lower Cholesky factor of a symmetric positive definite float64 device matrix, right-
    looking by blocks: small diagonal factorisations + one triangular solve per block column, and the
    trailing update as plain GEMMs (hipBLASLt fp64) over the LOWER block trapezoid only (the full
    square update did twice the flops).  The library potrf ran at ~4 TFLOP/s on 32k junctions.
    Only the lower triangle of the result is valid."""
    import torch
    m = A.shape[0]
    for k in range(0, m, bs):
        e = min(k + bs, m)
        A[k:e, k:e] = torch.linalg.cholesky(A[k:e, k:e])
        if e < m:
            L11 = A[k:e, k:e]
            A[e:, k:e] = torch.linalg.solve_triangular(L11, A[e:, k:e].T, upper=False).T     # L21 = A21 L11^-T
            L21 = A[e:, k:e]
            for j in range(e, m, 2 * bs):                      # block columns of the trailing matrix, rows from the diagonal down
                je = min(j + 2 * bs, m)
                A[j:, j:je].addmm_(L21[j - e:], L21[j - e:je - e].T, alpha=-1.0)
    return A


class BlockedCholeskySolve(object):
    """x = (L L^T)^-1 b for a dense lower factor L and 4 right-hand sides: the diagonal blocks are
    inverted once, after which forward and backward substitution are `cslam_chol_solve4_dev`
    (csrc/mac_kernels.hip): matrix x [bs][4] products that stream L exactly twice per solve at HBM
    speed (library potrs/trsm: ~0.1 s per call on 32k junctions; thin-right-hand-side GEMMs: 8 ms)."""

    def __init__(self, L, bs=2048):
        import torch
        self.L, self.bs, self.m = L, bs, L.shape[0]
        assert L.dim() == 2 and L.shape[0] == L.shape[1] and 1 in L.stride()
        # library factorisations come back column-major (element (r, c) at c * ld + r); the blocked one above is row-major
        self.col_major = int(L.stride(1) != 1)
        self.ld = L.stride(0) if not self.col_major else L.stride(1)
        self.lib = _lib.load()
        nb = (self.m + bs - 1) // bs
        self.dinv = torch.zeros((nb, bs, bs), dtype=L.dtype, device=L.device)
        for t in range(nb):
            k = t * bs
            e = min(k + bs, self.m)
            eye = torch.eye(e - k, dtype=L.dtype, device=L.device)
            self.dinv[t, :e - k, :e - k] = torch.linalg.solve_triangular(L[k:e, k:e], eye, upper=False)
        self.tmp = torch.empty((bs, 4), dtype=L.dtype, device=L.device)

    def solve(self, b):
        import torch
        assert b.shape == (self.m, 4) and b.dtype == torch.float64
        x = b.contiguous().clone()
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(self.lib.cslam_chol_solve4_dev(_p(self.L), self.m, max(self.ld, self.m), self.col_major, _p(self.dinv), self.bs,
                                                  _p(x), _p(self.tmp), st))
        return x


class ChainReducedSolverGPU(object):
    # junctions; above this the reduced system is solved on the host (sparse LU).  A dense float64
    # factor of 64k junctions is 33 GB -- small next to 288 GB of HBM -- while a sparse LU of a
    # loop-closure graph with random long-range edges fills in almost completely.
    MAX_DENSE = 64000

    def __init__(self, L, ground, device="cuda"):
        import os, time, torch
        self.torch = torch
        _t = [time.perf_counter()]
        def _lap(tag):
            if os.environ.get('CSLAM_MAC_TIMING'):
                torch.cuda.synchronize(); _t.append(time.perf_counter()); print(f'      [{tag} {(_t[-1]-_t[-2])*1e3:.0f} ms]', end='', flush=True)
        self.lib = _lib.load()
        host = ChainReducedSolver.__new__(ChainReducedSolver)
        ChainReducedSolver.__init__(host, L, ground, factorize=False)
        _lap('host structure')
        self.host = host
        n, nJ = host.n, host.nJ
        self.n, self.nJ = n, nJ
        dev = torch.device(device)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        self.is_j = t(host.is_j, np.uint8)
        self.r = t(host.r if n > 1 else np.zeros(1), np.float64)
        self.J = t(host.J, np.int64)
        nseg = len(host.sa)
        seg_start_of = np.full(nJ, -1, dtype=np.int32); seg_end_of = np.full(nJ, -1, dtype=np.int32)
        seg_start_of[host.jid[host.sa]] = np.arange(nseg, dtype=np.int32)
        seg_end_of[host.jid[host.sb]] = np.arange(nseg, dtype=np.int32)
        self.seg_start_of, self.seg_end_of = t(seg_start_of, np.int32), t(seg_end_of, np.int32)
        self.sa, self.sb = t(host.sa if nseg else np.zeros(1), np.int64), t(host.sb if nseg else np.zeros(1), np.int64)
        self.Rl = t(host.Rl if nseg else np.ones(1), np.float64)
        self.Rn = t(host.R, np.float64)
        self.jid = t(host.jid, np.int32)
        self.seg_of = t(host.seg_of_start[host.start], np.int32)
        nch = (n + 2047) // 2048
        f64 = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
        self.Bn, self.Qn, self.tmp = f64(n, 4), f64(n, 4), f64(n, 4)
        self.scratch = f64(9 * nch + 64)
        self.bt = f64(nJ, 4)
        self.free = t(host.free, np.int64)
        _lap('upload')
        self.dense = nJ - 1 <= self.MAX_DENSE
        if self.dense:
            # assemble the grounded junction Laplacian directly in HBM from the reduced edge list
            # (a 40k-junction matrix is 12.8 GB: never materialised on the host)
            g = int(host.jid[host.ground])
            m = nJ - 1
            Sf = torch.zeros((m, m), dtype=torch.float64, device=dev)
            ri, rj, rw = t(host.red_i, np.int64), t(host.red_j, np.int64), t(host.red_w, np.float64)
            fi = ri - (ri > g).to(torch.int64)
            fj = rj - (rj > g).to(torch.int64)
            ki, kj = ri != g, rj != g
            both = ki & kj
            Sf.index_put_((fi[ki], fi[ki]), rw[ki], accumulate=True)
            Sf.index_put_((fj[kj], fj[kj]), rw[kj], accumulate=True)
            Sf.index_put_((fi[both], fj[both]), -rw[both], accumulate=True)
            Sf.index_put_((fj[both], fi[both]), -rw[both], accumulate=True)
            _lap('assemble nJ=%d' % nJ)
            if m > 4096 and os.environ.get('CSLAM_MAC_CHOL', 'blocked') == 'blocked':
                self.chol = blocked_cholesky_(Sf)              # in place: the upper triangle keeps stale values, never read
            else:
                self.chol = torch.linalg.cholesky(Sf)
            del Sf
            _lap('cholesky')
            self.tri = BlockedCholeskySolve(self.chol, 2048 if m > 4096 else 512) if m >= 64 else None
        else:
            host.factorize()
        _lap('block inverses')
        if os.environ.get('CSLAM_MAC_TIMING'):
            print(flush=True)

    def solve(self, X):
        """X [n,4] float64 device tensor -> A^-1 X (row `ground` = 0)."""
        torch = self.torch
        assert X.shape == (self.n, 4) and X.dtype == torch.float64 and X.is_contiguous()
        st = C.c_void_p(torch.cuda.current_stream(X.device).cuda_stream)
        _lib.check(self.lib.cslam_chain_forward_dev(
            _p(X), _p(self.is_j), _p(self.r), self.n, _p(self.J), self.nJ, _p(self.seg_start_of),
            _p(self.seg_end_of), _p(self.sa), _p(self.sb), _p(self.Rl), _p(self.Bn), _p(self.Qn), _p(self.tmp),
            _p(self.scratch), _p(self.bt), st))
        xJ = torch.zeros((self.nJ, 4), dtype=torch.float64, device=X.device)
        if self.nJ > 1:
            rhs = self.bt[self.free]
            if self.dense:
                xJ[self.free] = self.tri.solve(rhs) if self.tri is not None else torch.cholesky_solve(rhs, self.chol)
            else:
                xJ[self.free] = torch.from_numpy(self.host.lu.solve(rhs.cpu().numpy())).to(X.device)
        out = torch.empty_like(X)
        _lib.check(self.lib.cslam_chain_backward_dev(
            _p(xJ), _p(self.Bn), _p(self.Qn), _p(self.r), _p(self.Rn), _p(self.jid), _p(self.seg_of), _p(self.sa),
            _p(self.sb), _p(self.Rl), self.n, _p(out), st))
        return out


_START_CACHE = {}


def fiedler_tracemin_chain_gpu(L, tol=1e-8, seed=None, device="cuda", stats=None):
    """TraceMIN-Fiedler (same start block, projection and stopping rule as the reference's
    networkx call, see fiedler.py) with every O(n) step on the GPU.  Returns (lambda_2, v numpy)."""
    import torch
    if seed is None:
        seed = np.random.RandomState(7)
    L = sp.csr_matrix(L, dtype=np.float64)
    L.sort_indices()
    n = L.shape[0]
    assert n > 4, "use the host solver for tiny graphs"
    dev = torch.device(device)
    lib = _lib.load()
    # the reference re-seeds RandomState(7) for every call (mac.py:56-58): the start block only depends on n
    key = (n, str(dev), seed.get_state()[1].tobytes())
    if _START_CACHE.get('key') != key:
        _START_CACHE['key'] = key
        _START_CACHE['X0'] = torch.from_numpy(np.ascontiguousarray(np.asarray(seed.normal(size=(4, n))).T)).to(dev)
    X = _START_CACHE['X0'].clone()
    ground = int((L.indptr[1:] - L.indptr[:-1]).argmax())
    solver = ChainReducedSolverGPU(L, ground, device)
    indptr = torch.from_numpy(L.indptr.astype(np.int64)).to(dev)
    indices = torch.from_numpy(L.indices.astype(np.int32)).to(dev)
    data = torch.from_numpy(L.data).to(dev)
    Lnorm = float(abs(L).sum(axis=1).max())
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    W = torch.empty_like(X)
    partial = torch.empty(20 * 1024, dtype=torch.float64, device=dev)
    out20 = torch.empty(20, dtype=torch.float64, device=dev)
    small = torch.empty(16 + 4, dtype=torch.float64, device=dev)       # 4x4 matrix + 4-vector staging

    def gram(A, B):
        """(A^T B as numpy 4x4, column sums of B)"""
        _lib.check(lib.cslam_block4_gram_dev(_p(A), _p(B), n, _p(partial), _p(out20), st))
        h = out20.cpu().numpy()
        return h[:16].reshape(4, 4).copy(), h[16:].copy()

    def affine(A, M, shift=None):
        """A @ M - shift, one streaming pass"""
        small[:16] = torch.from_numpy(np.ascontiguousarray(M, dtype=np.float64).reshape(-1)).to(dev)
        sp_ = None
        if shift is not None:
            small[16:] = torch.from_numpy(np.ascontiguousarray(shift, dtype=np.float64)).to(dev)
            sp_ = C.c_void_p(small.data_ptr() + 16 * 8)
        out = torch.empty_like(A)
        _lib.check(lib.cslam_block4_affine_dev(_p(A), n, _p(small), sp_, _p(out), st))
        return out

    def orthonormalise(X):
        for _ in range(2):                                   # CholQR2: X = Q R
            G, _ = gram(X, X)
            R = np.linalg.cholesky(G).T
            X = affine(X, np.linalg.inv(R))
        return X

    import time
    import scipy.linalg
    if stats is not None:
        torch.cuda.synchronize(); stats['setup_s'] = time.perf_counter() - stats.get('t0', time.perf_counter()); stats['iters'] = 0
        stats['nJ'] = solver.nJ; t_loop = time.perf_counter()
    _, cs = gram(X, X)
    X = affine(X, np.eye(4), cs / n)                         # project out the constant vector
    while True:
        if stats is not None:
            stats['iters'] += 1
        X = orthonormalise(X)
        _lib.check(lib.cslam_csr_spmm4_dev(_p(indptr), _p(indices), _p(data), n, _p(X), _p(W), st))
        H, _ = gram(X, W)
        sigma, Y = scipy.linalg.eigh(0.5 * (H + H.T))
        X = affine(X, Y)
        small[:4] = torch.from_numpy(np.ascontiguousarray(Y[:, 0])).to(dev)
        _lib.check(lib.cslam_block4_residual_dev(_p(W), _p(X), n, _p(small), float(sigma[0]), _p(partial), _p(out20), st))
        res = float(out20[0].item()) / Lnorm
        if res < tol:
            break
        Wi = solver.solve(X)
        M, _ = gram(Wi, X)                                   # W^T X
        X = affine(Wi, np.linalg.inv(M).T)
        _, cs = gram(X, X)
        X = affine(X, np.eye(4), cs / n)
    if stats is not None:
        torch.cuda.synchronize(); stats['loop_s'] = time.perf_counter() - t_loop
    return float(sigma[0]), X[:, 0].cpu().numpy()
