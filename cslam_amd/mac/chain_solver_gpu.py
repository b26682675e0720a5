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
            # L21 = A21 L11^-T as a right-hand solve on the stored layouts (no transposed copies of the 30k x 2048 panel)
            A[e:, k:e] = torch.linalg.solve_triangular(L11.T, A[e:, k:e], upper=True, left=False)
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
        self.dinvT = self.dinv.transpose(1, 2).contiguous()
        self.tmp = torch.empty((bs, 4), dtype=L.dtype, device=L.device)

    def solve(self, b):
        import torch
        assert b.shape == (self.m, 4) and b.dtype == torch.float64
        x = b.contiguous().clone()
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(self.lib.cslam_chol_solve4_dev(_p(self.L), self.m, max(self.ld, self.m), self.col_major, _p(self.dinv), _p(self.dinvT),
                                                  self.bs, _p(x), _p(self.tmp), st))
        return x


class _DensePool(object):
    """Float64 device buffers for the dense junction matrix, handed from one solver to the next: MAC builds a new, slightly
    larger system in every Frank-Wolfe iteration, and a fresh multi-gigabyte hipMalloc / hipFree per system cost up to 0.1 s."""
    free = []

    @classmethod
    def take(cls, numel, device):
        import torch
        best = None
        for t in cls.free:
            if t.device == device and t.numel() >= numel and (best is None or t.numel() < best.numel()):
                best = t
        if best is not None:
            cls.free.remove(best)
            return best
        cls.free[:] = [t for t in cls.free if t.device != device]          # too small: let them go before growing
        return torch.empty(int(numel * 1.3) + 1024, dtype=torch.float64, device=device)

    @classmethod
    def give(cls, t):
        if t is not None and len(cls.free) < 2:
            cls.free.append(t)


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
            self._pool_buf = _DensePool.take(m * m, dev)
            Sf = self._pool_buf[:m * m].view(m, m)
            Sf.zero_()
            ri, rj, rw = t(host.red_i, np.int64), t(host.red_j, np.int64), t(host.red_w, np.float64)
            fi = ri - (ri > g).to(torch.int64)
            fj = rj - (rj > g).to(torch.int64)
            ki, kj = ri != g, rj != g
            both = ki & kj
            # every entry's terms are added in a FIXED order (stable sort by entry, sequential sum per entry): scattering with
            # atomic adds left the order of the ~3 terms of an entry to the scheduler, a 1e-16 run-to-run jitter in the matrix
            ii = torch.cat((fi[ki], fj[kj], fi[both], fj[both]))
            jj = torch.cat((fi[ki], fj[kj], fj[both], fi[both]))
            vv = torch.cat((rw[ki], rw[kj], -rw[both], -rw[both]))
            key, order = torch.sort(ii * m + jj, stable=True)
            ukey, counts = torch.unique_consecutive(key, return_counts=True)
            sums = torch.segment_reduce(vv[order], "sum", lengths=counts)
            Sf.view(-1)[ukey] = sums
            _lap('assemble nJ=%d' % nJ)
            if m > 4096:                                           # own blocked factorisation: 0.32 s at 32k junctions, library potrf 0.47 s
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

    def __del__(self):
        try:
            _DensePool.give(getattr(self, '_pool_buf', None))
        except Exception:
            pass

    def solve(self, X, out=None):
        """X [n,4] float64 device tensor -> A^-1 X (row `ground` = 0), into `out` when given."""
        torch = self.torch
        assert X.shape == (self.n, 4) and X.dtype == torch.float64 and X.is_contiguous()
        st = C.c_void_p(torch.cuda.current_stream(X.device).cuda_stream)
        _lib.check(self.lib.cslam_chain_forward_dev(
            _p(X), _p(self.is_j), _p(self.r), self.n, _p(self.J), self.nJ, _p(self.seg_start_of),
            _p(self.seg_end_of), _p(self.sa), _p(self.sb), _p(self.Rl), _p(self.Bn), _p(self.Qn), _p(self.tmp),
            _p(self.scratch), _p(self.bt), st))
        if getattr(self, '_xJ', None) is None:
            self._xJ = torch.zeros((self.nJ, 4), dtype=torch.float64, device=X.device)     # the grounded row stays 0
        xJ = self._xJ
        if self.nJ > 1:
            rhs = self.bt[self.free]
            if self.dense:
                xJ[self.free] = self.tri.solve(rhs) if self.tri is not None else torch.cholesky_solve(rhs, self.chol)
            else:
                xJ[self.free] = torch.from_numpy(self.host.lu.solve(rhs.cpu().numpy())).to(X.device)
        if out is None:
            out = torch.empty_like(X)
        _lib.check(self.lib.cslam_chain_backward_dev(
            _p(xJ), _p(self.Bn), _p(self.Qn), _p(self.r), _p(self.Rn), _p(self.jid), _p(self.seg_of), _p(self.sa),
            _p(self.sb), _p(self.Rl), self.n, _p(out), st))
        return out


_START_CACHE = {}


def fiedler_tracemin_chain_gpu(L, tol=1e-8, seed=None, device="cuda", stats=None):
    """TraceMIN-Fiedler (same start block, projection and stopping rule as the reference's
    networkx call, see fiedler.py) with every O(n) step on the GPU.  Returns (lambda_2, v numpy)."""
    import os
    import time as _time
    import torch
    _tt = [_time.perf_counter()]

    def _lap(tag):
        if os.environ.get('CSLAM_MAC_TIMING') == '2':
            torch.cuda.synchronize(); _tt.append(_time.perf_counter()); print(f'      <{tag} {(_tt[-1] - _tt[-2]) * 1e3:.0f} ms>', end='', flush=True)
    if seed is None:
        seed = np.random.RandomState(7)
    L = sp.csr_matrix(L, dtype=np.float64)
    L.sort_indices()
    _lap('csr')
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
    _lap('start block')
    solver = ChainReducedSolverGPU(L, ground, device)
    _lap('solver')
    indptr = torch.from_numpy(L.indptr.astype(np.int64)).to(dev)
    indices = torch.from_numpy(L.indices.astype(np.int32)).to(dev)
    data = torch.from_numpy(L.data).to(dev)
    _lap('csr upload')
    # max row sum of |L| (the stopping rule's scale): segment sums over the CSR arrays (`abs(L).sum(axis=1)` builds two
    # temporaries of the matrix); an empty row would pick up its successor's first entry, so those are masked
    if L.nnz:
        starts = np.minimum(L.indptr[:-1], L.nnz - 1)
        rs = np.add.reduceat(np.abs(L.data), starts)
        rs[L.indptr[1:] == L.indptr[:-1]] = 0.0
        Lnorm = float(rs.max())
    else:
        Lnorm = 0.0
    _lap('norm')
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    W = torch.empty_like(X)
    partial = torch.empty(20 * 1024, dtype=torch.float64, device=dev)
    out20 = torch.empty(20, dtype=torch.float64, device=dev)
    # The 4 x 4 algebra between the streaming passes stays on the host (LAPACK, as in the reference); its operands travel as
    # kernel arguments and its inputs come back as one 160-byte copy per pass (`cslam_block4_*_sync / _host`): no torch
    # tensors are created inside the loop, X ping-pongs between two resident buffers.
    h20 = np.empty(20, dtype=np.float64)
    h20_p = h20.ctypes.data_as(C.c_void_p)
    pool = [torch.empty_like(X), torch.empty_like(X)]         # free [n,4] buffers (X itself is the third)

    def gram(A, B):
        """(A^T B as numpy 4x4, column sums of B)"""
        _lib.check(lib.cslam_block4_gram_sync(_p(A), _p(B), n, _p(partial), _p(out20), h20_p, st))
        return h20[:16].reshape(4, 4).copy(), h20[16:].copy()

    def affine_into(out, A, M, shift=None):
        """out = A @ M - shift (one streaming pass; out must not be A)"""
        Mc = np.ascontiguousarray(M, dtype=np.float64)
        sc_ = None if shift is None else np.ascontiguousarray(shift, dtype=np.float64)
        _lib.check(lib.cslam_block4_affine_host(_p(A), n, Mc.ctypes.data_as(C.c_void_p),
                                                None if sc_ is None else sc_.ctypes.data_as(C.c_void_p), _p(out), st))
        return out

    def affine(A, M, shift=None):
        """A @ M - shift into a free buffer; A's buffer becomes free"""
        out = affine_into(pool.pop(), A, M, shift)
        pool.append(A)
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
    h1 = np.empty(1, dtype=np.float64)
    while True:
        if stats is not None:
            stats['iters'] += 1
        X = orthonormalise(X)
        _lib.check(lib.cslam_csr_spmm4_dev(_p(indptr), _p(indices), _p(data), n, _p(X), _p(W), st))
        H, _ = gram(X, W)
        sigma, Y = scipy.linalg.eigh(0.5 * (H + H.T))
        X = affine(X, Y)
        y0 = np.ascontiguousarray(Y[:, 0], dtype=np.float64)
        _lib.check(lib.cslam_block4_residual_sync(_p(W), _p(X), n, y0.ctypes.data_as(C.c_void_p), float(sigma[0]), _p(partial),
                                                  _p(out20), h1.ctypes.data_as(C.c_void_p), st))
        res = float(h1[0]) / Lnorm
        if res < tol:
            break
        Wi = solver.solve(X, out=pool.pop())
        # X <- Wi (Wi^T X)^-1, then minus its column means: the column sums of the product are those of Wi times the same
        # matrix, so one pass over (X, Wi) gives both and one pass writes the projected block (over the old X)
        Mt, csW = gram(X, Wi)                                # X^T Wi = (Wi^T X)^T and the column sums of Wi
        Minv_T = np.linalg.inv(Mt.T).T
        affine_into(X, Wi, Minv_T, (csW @ Minv_T) / n)
        pool.append(Wi)
    if stats is not None:
        torch.cuda.synchronize(); stats['loop_s'] = time.perf_counter() - t_loop
    return float(sigma[0]), X[:, 0].cpu().numpy()


def fiedler_tracemin_hip(L, tol=1e-8, seed=7, device="cuda", stats=None, x0=None, max_iters=0):
    """The same computation through the C ABI's one-call entry point `cslam_fiedler` (csrc/fiedler.hip): chain / junction
    structure, dense factor (rocBLAS / rocSOLVER), TraceMIN loop and its 4 x 4 algebra all in native code -- what a host
    without Python calls in place of cslam/mac/mac.py:35-59.  `seed` is the integer the reference seeds RandomState with
    (7, mac.py:56-58).  Returns (lambda_2, v numpy)."""
    import time
    import torch
    L = sp.csr_matrix(L, dtype=np.float64)
    L.sum_duplicates()
    L.sort_indices()
    n = L.shape[0]
    lib = _lib.load()
    indptr = np.ascontiguousarray(L.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(L.indices, dtype=np.int32)
    data = np.ascontiguousarray(L.data, dtype=np.float64)
    lam, iters, v = C.c_double(0.0), C.c_int(0), np.empty(n, dtype=np.float64)
    x0p = None if x0 is None else np.ascontiguousarray(x0, dtype=np.float64)
    assert x0p is None or x0p.shape == (n, 4)
    dev = torch.device(device)
    t0 = time.perf_counter()
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.cslam_fiedler(n, indptr.ctypes.data_as(C.c_void_p), indices.ctypes.data_as(C.c_void_p),
                                     data.ctypes.data_as(C.c_void_p), None if x0p is None else x0p.ctypes.data_as(C.c_void_p),
                                     int(seed), float(tol), int(max_iters), C.byref(lam), v.ctypes.data_as(C.c_void_p),
                                     C.byref(iters), st))
    if stats is not None:
        stats['iters'] = iters.value
        stats['total_s'] = time.perf_counter() - t0
    return float(lam.value), v
