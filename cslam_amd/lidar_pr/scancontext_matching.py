#!/usr/bin/env python
"""ScanContext matching on an MI355X.

Drop-in for the reference class of the same name (cslam/lidar_pr/scancontext_matching.py:5-104):
same constructor defaults, `add_item`, `search`, `search_best`, and the public attributes
`shape`, `num_candidates`, `threshold`, `items`, `nb_items`, `scancontexts`, `ringkeys`.
Scan contexts, ring keys and column norms live in HBM behind libcslam_hip.so; the search is
hand-written HIP (csrc/scancontext.hip): brute-force ring-key k-NN in place of the per-query
KD-tree rebuild, then one workgroup per (query, candidate) for the yaw-shifted column-cosine
distance.  No CPU path: without the library or a GPU every call raises `CslamHipError`.

Batch extensions used by LoopClosureSparseMatching's batched callers: `add_items`, `search_batch`
(same return convention as NearestNeighborsMatching.search_batch, k fixed to the reference's 1).
"""
import ctypes as C

import numpy as np

from .. import _lib


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class ScanContextMatching(object):
    """Nearest Neighbor matching of description vectors
    """

    def __init__(self, shape=[20, 60], num_candidates=10, threshold=0.15, device=0):
        """ Initialization
            Default configs are the same as in the original paper
        """
        self.shape = shape
        self.num_candidates = num_candidates
        self.threshold = threshold
        self.items = dict()
        self.nb_items = 0
        self.device = device
        self.last_yaw_diff_deg = None
        _lib.require_gpu()
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.cslam_scbank_create(device, int(shape[0]), int(shape[1]), 1000, C.byref(h)))
        self._bank = h
        self._len = int(shape[0]) * int(shape[1])

    def __del__(self):
        try:
            if getattr(self, "_bank", None) is not None:
                self._lib.cslam_scbank_destroy(self._bank)
                self._bank = None
        except Exception:
            pass

    # `n` is what the batched callers of LoopClosureSparseMatching read on either matcher type
    @property
    def n(self):
        return self.nb_items

    def _capacity(self):
        cap = 1000
        while cap < self.nb_items:
            cap *= 2
        return cap

    def _read(self):
        cap = self._capacity()
        sc = np.zeros((cap, self.shape[0], self.shape[1]))
        rk = np.zeros((cap, self.shape[0]))
        if self.nb_items:
            _lib.check(self._lib.cslam_scbank_read_host(self._bank, 0, self.nb_items, _vp(sc), _vp(rk)))
        return sc, rk

    @property
    def scancontexts(self):
        """Host copy, shape (capacity, rings, sectors) float64, zeros past nb_items
        (the reference's public array, scancontext_matching.py:18)."""
        return self._read()[0]

    @property
    def ringkeys(self):
        """Host copy, shape (capacity, rings) float64 (scancontext_matching.py:19)."""
        return self._read()[1]

    # ------------------------------------------------------- reference API ----
    def add_item(self, descriptor, item):
        """Add item to the matching list (scancontext_matching.py:23-42)

        Args:
            descriptor (np.array): descriptor
            item: identification info (e.g., int)
        """
        sc = np.ascontiguousarray(np.asarray(descriptor).reshape(self.shape), dtype=np.float64)
        _lib.check(self._lib.cslam_scbank_add_host(self._bank, _vp(sc), 1))
        self.items[self.nb_items] = item
        self.nb_items = self.nb_items + 1

    def search(self, query, k):
        """Search for nearest neighbors (scancontext_matching.py:44-87)

        Args:
            query (np.array): descriptor to match
            k (int): number of best matches to return (as in the reference, only the best one is)

        Returns:
            list(int, np.array): best matches
        """
        if self.nb_items < 1:
            return [None], [None]
        rows, sims, _, yaw = self._search(np.asarray(query).reshape([1] + list(self.shape)), None)
        # as in the reference, "no candidate closer than 1.0" answers the first item with 0.0
        self.last_yaw_diff_deg = float(yaw[0]) * (360 / self.shape[1]) if rows[0] >= 0 else 0
        nn_idx = int(rows[0]) if rows[0] >= 0 else 0
        return [self.items[nn_idx]], [float(sims[0])]

    def search_best(self, query):
        """Search for the nearest neighbor
            Implementation for compatibily only (scancontext_matching.py:89-104)

        Returns:
            int, np.array: best match
        """
        if self.nb_items < 1:
            return None, None
        idxs, sims = self.search(query, 1)
        return idxs[0], sims[0]

    # ---------------------------------------------------------- extensions ----
    def _search(self, queries, row_limit, diagnostics=False):
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, self._len)
        nq = q.shape[0]
        rows = np.empty(nq, dtype=np.int64)
        sims = np.empty(nq, dtype=np.float64)
        yaw = np.empty(nq, dtype=np.int32)
        lim = None
        if row_limit is not None:
            lim = np.ascontiguousarray(row_limit, dtype=np.int64)
            assert lim.shape == (nq,)
        cand = cdist = cyaw = None
        if diagnostics:
            cand = np.empty((nq, self.num_candidates), dtype=np.int64)
            cdist = np.empty((nq, self.num_candidates), dtype=np.float64)
            cyaw = np.empty((nq, self.num_candidates), dtype=np.int32)
        _lib.check(self._lib.cslam_scbank_search_host(self._bank, _vp(q), nq, int(self.num_candidates), _vp(lim),
                                                      _vp(rows), _vp(sims), _vp(yaw), _vp(cand), _vp(cdist),
                                                      _vp(cyaw)))
        return rows, sims, (cand, cdist, cyaw), yaw

    def add_items(self, descriptors, items):
        """Append m scan contexts at once ([m, rings*sectors] or [m, rings, sectors])."""
        sc = np.ascontiguousarray(descriptors, dtype=np.float64).reshape(-1, self._len)
        items = list(items)
        assert len(items) == sc.shape[0]
        _lib.check(self._lib.cslam_scbank_add_host(self._bank, _vp(sc), sc.shape[0]))
        for it in items:
            self.items[self.nb_items] = it
            self.nb_items += 1

    def search_batch(self, queries, k=1, row_limit=None, mode=None):
        """Batch of `search` calls in one launch sequence.  Returns (rows [nq,1] int64, sims [nq,1]
        float64, cnt [nq] int32) with the reference's conventions folded in: a query that sees at
        least one bank row always has one answer (row 0 with similarity 0.0 when no candidate is
        closer than 1.0); a query that sees none has cnt 0."""
        q = np.asarray(queries)
        nq = q.shape[0]
        rows, sims, _, _ = self._search(q, row_limit)
        visible = np.full(nq, self.nb_items, dtype=np.int64) if row_limit is None else \
            np.minimum(np.asarray(row_limit, dtype=np.int64), self.nb_items)
        cnt = (visible > 0).astype(np.int32)
        rows = np.where(rows >= 0, rows, 0)
        rows = np.where(cnt > 0, rows, -1)
        return rows[:, None], sims[:, None], cnt

    def search_diagnostics(self, queries, row_limit=None):
        """Stage-level outputs for tests: dict(best_idx, best_sim, best_yaw, cand, cdist, cyaw)."""
        rows, sims, (cand, cdist, cyaw), yaw = self._search(np.asarray(queries), row_limit, diagnostics=True)
        return dict(best_idx=rows, best_sim=sims, best_yaw=yaw, cand=cand, cdist=cdist, cyaw=cyaw)
