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
