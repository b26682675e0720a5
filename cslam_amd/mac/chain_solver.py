"""Exact solves with a pose-graph Laplacian by eliminating the odometry chains in closed form.

The Laplacians MAC hands to the Fiedler solver (cslam/mac/mac.py:61-77) are, by construction
(cslam/algebraic_connectivity_maximization.py:348-362), long odometry CHAINS (edges i -- i+1) plus a
comparatively small set of loop-closure edges.  The reference solves with them through a general
sparse LU (networkx `_tracemin_fiedler` -> SciPy SuperLU, 85 % of MAC's time).  Here
every maximal run of chain nodes between two *junctions* (nodes touched by a loop edge, chain ends,
the grounded node) is eliminated exactly with prefix sums -- a path of conductances is a 1-D
resistor network:

    flow on edge e      f_e = f_1 + B_{e-1}          B = running sum of the injected right-hand side
    potential           x_k = x_a - f_1 R_k - Q_k    R = running resistance, Q = running sum of r_e B_{e-1}
    boundary condition  f_1 = (x_a - x_b - Q_l) / R_l

which leaves a small Laplacian on the junctions only (segment conductance 1/R_l + loop edges, right-hand
side corrected by Q_l/R_l and B_{l-1}), solved with a sparse factorisation, followed by the
closed-form back-substitution of the interior potentials.  The O(n) part is three segmented scans
and two gathers per right-hand side: HBM-streaming work with device twins in
cslam_amd/csrc/mac_kernels.hip; the numpy code below is the host statement of the same arithmetic
(used when no GPU array is given, and as the oracle of the kernels).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


class ChainReducedSolver:
    """x = A^-1 b where A = L with node `ground` clamped to 0 (networkx's A[g,g] = inf)."""

    def __init__(self, L, ground, factorize=True):
        L = sp.csr_matrix(L, dtype=np.float64)
        n = L.shape[0]
        self.n, self.ground = n, int(ground)
        # chain conductances c[i] of edge (i, i+1) and the loop edges |i - j| >= 2, straight from the CSR arrays (row-major
        # order, as `sp.triu(L, 2).tocoo()` / `L.diagonal(1)` gave them: those two calls were half of this constructor)
        L.sum_duplicates()                                  # no-op for the canonical matrices MAC builds
        rows = np.repeat(np.arange(n, dtype=L.indices.dtype), np.diff(L.indptr))
        d = L.indices - rows                                # column - row of every stored entry
        c = np.zeros(max(n - 1, 0))
        if n > 1:
            sup = np.flatnonzero(d == 1)
            c[rows[sup]] = -L.data[sup]
        up = np.flatnonzero(d >= 2)
        li, lj, lw = rows[up].astype(np.int64), L.indices[up].astype(np.int64), -L.data[up]
        keep = lw != 0
        li, lj, lw = li[keep], lj[keep], lw[keep]
        is_j = np.zeros(n, dtype=bool)
        is_j[li] = True; is_j[lj] = True
        is_j[self.ground] = True
        is_j[0] = True; is_j[n - 1] = True
        broken = np.nonzero(c <= 0)[0]                    # missing chain edge: both ends are junctions
        is_j[broken] = True; is_j[broken + 1] = True
        J = np.nonzero(is_j)[0]
        self.J, self.nJ = J, len(J)
        jid = np.full(n, -1, dtype=np.int64); jid[J] = np.arange(self.nJ)
        # per node: the junction at or before it (start of its segment)
        start = np.maximum.accumulate(np.where(is_j, np.arange(n), -1))
        self.start, self.is_j = start, is_j
        r = np.zeros(max(n - 1, 0))
        pos = c > 0
        r[pos] = 1.0 / c[pos]
        self.r = r
        rc = np.concatenate([[0.0], np.cumsum(r)])        # rc[k] = sum r[0..k-1]
        self.R = rc[np.arange(n)] - rc[start]             # resistance from the segment start to node k
        # segments: consecutive junctions joined by a chain
        a, b = J[:-1], J[1:]
        seg = c[a] > 0 if n > 1 else np.zeros(0, dtype=bool)
        self.sa, self.sb = a[seg], b[seg]
        self.rc = rc
        self.Rl = rc[self.sb] - rc[self.sa]              # total resistance of every segment
        self.seg_of_start = np.full(n, -1, dtype=np.int64)
        self.seg_of_start[self.sa] = np.arange(len(self.sa))
        # reduced Laplacian on the junctions
        ri = np.concatenate([jid[self.sa], jid[li]])
        rj = np.concatenate([jid[self.sb], jid[lj]])
        rw = np.concatenate([1.0 / self.Rl, lw])
        self.red_i, self.red_j, self.red_w = ri, rj, rw        # reduced graph as an edge list (junction ids)
        S = sp.coo_matrix((np.concatenate([rw, rw, -rw, -rw]),
                           (np.concatenate([ri, rj, ri, rj]), np.concatenate([ri, rj, rj, ri]))),
                          shape=(self.nJ, self.nJ)).tocsc()
        g = jid[self.ground]
        self.free = np.concatenate([np.arange(g), np.arange(g + 1, self.nJ)])
        self.jid = jid
        self.Sf = S[self.free][:, self.free].tocsc()
        self.lu = None
        if factorize:
            self.factorize()

    def factorize(self):
        self.lu = spla.splu(self.Sf) if self.Sf.shape[0] > 0 else None

    def solve(self, Bm):
        """Bm [n] or [n, q] -> X of the same shape with X[ground] = 0."""
        Bm = np.asarray(Bm, dtype=np.float64)
        one = Bm.ndim == 1
        if one:
            Bm = Bm[:, None]
        n, q = Bm.shape
        bz = np.where(self.is_j[:, None], 0.0, Bm)
        cs = np.cumsum(bz, axis=0)
        Bn = cs - cs[self.start]                          # B at node k (0 at junctions)
        if n > 1:
            qc = np.concatenate([np.zeros((1, q)), np.cumsum(self.r[:, None] * Bn[:-1], axis=0)])
        else:
            qc = np.zeros((1, q))
        Q = qc[np.arange(n)] - qc[self.start]
        # reduced right-hand side
        bt = Bm[self.J].copy()
        Ql, Bl = qc[self.sb] - qc[self.sa], Bn[self.sb - 1]
        corr = Ql / self.Rl[:, None]
        np.add.at(bt, self.jid[self.sa], corr)
        np.add.at(bt, self.jid[self.sb], Bl - corr)
        xJ = np.zeros((self.nJ, q))
        if self.lu is not None:
            xJ[self.free] = self.lu.solve(np.ascontiguousarray(bt[self.free]))
        # back-substitution along the segments
        X = np.zeros((n, q))
        X[self.J] = xJ
        sidx = self.seg_of_start[self.start]              # segment of every node (-1: none)
        inner = (~self.is_j) & (sidx >= 0)
        k = np.nonzero(inner)[0]
        s = sidx[k]
        xa, xb = xJ[self.jid[self.sa[s]]], xJ[self.jid[self.sb[s]]]
        f1 = (xa - xb - Ql[s]) / self.Rl[s][:, None]
        X[k] = xa - f1 * self.R[k][:, None] - Q[k]
        return X[:, 0] if one else X


def fiedler_tracemin_chain(L, tol=1e-8, seed=None, solver_cls=ChainReducedSolver):
    """Same TraceMIN iteration as fiedler.fiedler_tracemin_lu (start block, projection, stopping
    rule), with the inner solves done by the chain-reduced solver instead of a full sparse LU."""
    import scipy.linalg
    if seed is None:
        seed = np.random.RandomState(7)
    L = sp.csr_matrix(L, dtype=np.float64)
    n = L.shape[0]
    q = min(4, n - 1)
    X = np.asarray(seed.normal(size=(q, n))).T
    ground = int((L.indptr[1:] - L.indptr[:-1]).argmax())
    solver = solver_cls(L, ground)
    Lnorm = abs(L).sum(axis=1).flatten().max()
    X -= X.sum(axis=0) / n
    while True:
        X = np.linalg.qr(X)[0]
        W = L @ X
        H = X.T @ W
        sigma, Y = scipy.linalg.eigh(H)
        X = X @ Y
        res = np.abs(W @ Y[:, 0] - sigma[0] * X[:, 0]).sum() / Lnorm
        if res < tol:
            break
        W = solver.solve(X)
        X = (scipy.linalg.inv(W.T @ X) @ W.T).T
        X -= X.sum(axis=0) / n
    return sigma[0], np.asarray(X)[:, 0]
