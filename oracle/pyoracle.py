"""ctypes wrapper of oracle/liboracle.so (TEST INFRASTRUCTURE; see nns_oracle.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_nns_search.restype = C.c_int
        _lib.oracle_nns_search.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int64,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def nns_search(bank, queries, k, row_limit=None):
    """Oracle for NearestNeighborsMatching.search over a batch (cslam/nns_matching.py:42-61).
    bank [n,d] float32; queries [nq,d] float32 or float64.
    Returns (rows [nq,k] int64, sims [nq,k] float64, cnt [nq] int32)."""
    lib = load()
    bank = np.ascontiguousarray(bank, dtype=np.float32)
    q = np.ascontiguousarray(queries)
    if q.dtype != np.float32:
        q = np.ascontiguousarray(q, dtype=np.float64)
    n, d = bank.shape
    nq = q.shape[0]
    assert q.shape[1] == d
    idx = np.empty((nq, k), dtype=np.int64)
    sims = np.empty((nq, k), dtype=np.float64)
    cnt = np.empty(nq, dtype=np.int32)
    lim = None
    if row_limit is not None:
        lim = np.ascontiguousarray(row_limit, dtype=np.int64)
    rc = lib.oracle_nns_search(bank.ctypes.data_as(C.c_void_p), n, d, q.ctypes.data_as(C.c_void_p),
                               int(q.dtype == np.float64), nq, k,
                               lim.ctypes.data_as(C.c_void_p) if lim is not None else None,
                               idx.ctypes.data_as(C.c_void_p), sims.ctypes.data_as(C.c_void_p),
                               cnt.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return idx, sims, cnt


# ---- lidar ScanContext (sc_oracle.c) -----------------------------------------------------
def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def sc_ringkey(sc):
    """Oracle for sc2rk (cslam/lidar_pr/scancontext_utils.py:78-79). sc [R,S] float64."""
    lib = load()
    sc = np.ascontiguousarray(sc, dtype=np.float64)
    rk = np.empty(sc.shape[0], dtype=np.float64)
    assert lib.oracle_sc_ringkey(_p(sc), sc.shape[0], sc.shape[1], _p(rk)) == 0
    return rk


def sc_distance(sc1, sc2):
    """Oracle for distance_sc (scancontext_utils.py:81-113) -> (dist, yaw_diff)."""
    lib = load()
    sc1 = np.ascontiguousarray(sc1, dtype=np.float64)
    sc2 = np.ascontiguousarray(sc2, dtype=np.float64)
    d = C.c_double()
    y = C.c_int()
    assert lib.oracle_sc_distance(_p(sc1), _p(sc2), sc1.shape[0], sc1.shape[1], C.byref(d), C.byref(y)) == 0
    return d.value, y.value


def sc_search(bank, queries, num_candidates=10, row_limit=None):
    """Oracle for ScanContextMatching.search over a batch (scancontext_matching.py:44-87).
    bank [n,R,S], queries [nq,R,S] float64.  Returns dict(best_idx [nq] (-1 = the reference's
    'no candidate under distance 1' branch), best_sim, best_yaw, cand [nq,C], cdist, cyaw)."""
    lib = load()
    bank = np.ascontiguousarray(bank, dtype=np.float64)
    q = np.ascontiguousarray(queries, dtype=np.float64)
    R, S = q.shape[1], q.shape[2]
    n, nq = bank.shape[0], q.shape[0]
    lim = np.ascontiguousarray(row_limit, dtype=np.int64) if row_limit is not None else None
    out = dict(best_idx=np.empty(nq, np.int64), best_sim=np.empty(nq, np.float64),
               best_yaw=np.empty(nq, np.int32), cand=np.empty((nq, num_candidates), np.int64),
               cdist=np.empty((nq, num_candidates), np.float64), cyaw=np.empty((nq, num_candidates), np.int32))
    lib.oracle_sc_search.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                     C.c_void_p] + [C.c_void_p] * 6
    rc = lib.oracle_sc_search(_p(bank), n, R, S, _p(q), nq, num_candidates, _p(lim), _p(out["best_idx"]),
                              _p(out["best_sim"]), _p(out["best_yaw"]), _p(out["cand"]), _p(out["cdist"]),
                              _p(out["cyaw"]))
    assert rc == 0
    return out


def ptcloud2sc(points, shape=(20, 60), max_length=80.0):
    """Oracle for ptcloud2sc (cslam/lidar_pr/scancontext_utils.py:46-75). points [n,3] -> [R,S] float64."""
    lib = load()
    pts = np.ascontiguousarray(points, dtype=np.float64)
    sc = np.empty(shape, dtype=np.float64)
    lib.oracle_ptcloud2sc.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_void_p]
    rc = lib.oracle_ptcloud2sc(_p(pts), pts.shape[0], shape[0], shape[1], float(max_length), _p(sc))
    if rc == -2:
        raise IndexError("sector index out of range (theta == 360), as in the reference")
    assert rc == 0
    return sc
