#!/usr/bin/env python
"""Nearest-neighbour matching of global descriptors on an MI355X.

Drop-in for the reference class of the same name (cslam/nns_matching.py:7-76):
same constructor, `add_item`, `search`, `search_best`, and the public attributes
`n`, `dim`, `items`, `data` the reference's callers and tests touch.  The bank
lives in HBM behind libcslam_hip.so; scores are computed by hand-written HIP
kernels (exact float64 scan for single queries, fp32-MFMA candidates + float64
re-score for batches).  There is no CPU path: without the library or a GPU every
call raises `CslamHipError`.

Batch / device-resident extensions (not in the reference; used by the batched
callers in this package and by bench.py): `add_items`, `search_batch`,
`add_items_device`, `search_device`.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import CslamHipError, MODE_AUTO, MODE_MFMA, MODE_SCAN  # noqa: F401


def _as_query_array(q):
    q = np.asarray(q)
    if q.dtype == np.float32:
        return np.ascontiguousarray(q), _lib.F32
    return np.ascontiguousarray(q, dtype=np.float64), _lib.F64


class NearestNeighborsMatching(object):
    """Nearest Neighbor matching of description vectors (HBM-resident bank)."""

    def __init__(self, dim=None, device=0):
        """Initialization

        Args:
            dim (int, optional): Global descriptor size. Defaults to None
                (inferred from the first vector, cslam/nns_matching.py:32-34).
            device (int): HIP device ordinal holding the bank.
        """
        self.n = 0
        self.dim = dim
        self.items = dict()
        self.device = device
        self._bank = None
        self._lib = None
        self._data_cache = (0, None)          # (rows already mirrored on the host, array) behind `.data`
        self._item_ids = (0, np.zeros(0, dtype=np.int64))     # (rows covered, int64 items) behind item_array()
        if dim is not None:
            self._create(dim)

    # ------------------------------------------------------------- plumbing ----
    def _create(self, dim):
        _lib.require_gpu()
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.cslam_bank_create(self.device, int(dim), 1000, C.byref(h)))
        self._bank = h
        self.dim = int(dim)

    def __del__(self):
        try:
            if self._bank is not None and self._lib is not None:
                self._lib.cslam_bank_destroy(self._bank)
                self._bank = None
        except Exception:
            pass

    @property
    def data(self):
        """Host view of the bank storage, shape (capacity, dim) float32, zeros past `n` (the reference's public
        array, cslam/nns_matching.py:21; [] before the first add).  The bank itself lives in HBM: the host copy is
        made on first access and kept until the next add (only the rows appended since are downloaded), so callers
        that touch `.data` freely, as the reference's tests do, do not pay a bank-sized transfer per access."""
        if self._bank is None:
            return []
        cap = 1000
        while cap < self.n:
            cap *= 2
        have, arr = self._data_cache
        if arr is None or arr.shape[0] != cap:
            new = np.zeros((cap, self.dim), dtype=np.float32)
            if arr is not None and have:
                new[:have] = arr[:have]
            arr = new
        if have < self.n:
            _lib.check(self._lib.cslam_bank_read_host(self._bank, have, self.n - have,
                                                      arr[have:self.n].ctypes.data_as(C.c_void_p)))
        self._data_cache = (self.n, arr)
        return arr

    # ------------------------------------------------------- reference API ----
    def add_item(self, vector, item):
        """Add item to the matching list (cslam/nns_matching.py:23-40)

        Args:
            vector (np.array): descriptor
            item: identification info (e.g., int)
        """
        vector = np.asarray(vector)
        assert vector.ndim == 1
        if self._bank is None:
            self._create(len(vector))
        if len(vector) != self.dim:
            raise ValueError(f"could not broadcast input array from shape ({len(vector)},) "
                             f"into shape ({self.dim},)")
        v, dt = _as_query_array(vector)
        _lib.check(self._lib.cslam_bank_add_host(self._bank, v.ctypes.data_as(C.c_void_p), dt, 1))
        self.items[self.n] = item
        self.n += 1

    def search(self, query, k):
        """Search for nearest neighbors (cslam/nns_matching.py:42-61)

        Args:
            query (np.array): descriptor to match
            k (int): number of best matches to return

        Returns:
            list(int, np.array): best matches
        """
        if self._bank is None:
            return [], []
        query = np.asarray(query)
        if query.ndim != 1 or len(query) != self.dim:
            raise ValueError(f"shapes ({query.shape}) and ({self.dim},) not aligned")
        if self.n == 0 or k <= 0:
            return [], np.zeros(0)
        idx, sims, cnt = self.search_batch(query[None, :], k)
        c = int(cnt[0])
        return [self.items[int(r)] for r in idx[0, :c]], sims[0, :c].copy()

    def search_best(self, query):
        """Search for the nearest neighbor (cslam/nns_matching.py:63-76)

        Returns:
            int, np.array: best match
        """
        if self._bank is None:
            return None, None
        items, similarities = self.search(query, 1)
        return items[0], similarities[0]

    # ---------------------------------------------------------- extensions ----
    def add_items(self, vectors, items):
        """Append many descriptors at once: vectors [m, dim], items iterable of m ids."""
        vectors = np.asarray(vectors)
        assert vectors.ndim == 2
        items = list(items)
        assert len(items) == vectors.shape[0]
        if vectors.shape[0] == 0:
            return
        if self._bank is None:
            self._create(vectors.shape[1])
        if vectors.shape[1] != self.dim:
            raise ValueError("descriptor dimension mismatch")
        v, dt = _as_query_array(vectors)
        _lib.check(self._lib.cslam_bank_add_host(self._bank, v.ctypes.data_as(C.c_void_p), dt, v.shape[0]))
        for it in items:
            self.items[self.n] = it
            self.n += 1

    def search_batch(self, queries, k, row_limit=None, mode=MODE_AUTO):
        """Top-k of every query row against the bank.

        row_limit: optional int64 [nq]; query j only sees bank rows < row_limit[j]
        (the causal order of global_descriptor_loop_closure_detection.py:157-160).
        Returns (rows [nq,k] int64 (-1 padded), sims [nq,k] float64 (NaN padded), cnt [nq] int32).
        """
        if self._bank is None:
            raise CslamHipError("search_batch on a bank that was never populated")
        q, dt = _as_query_array(queries)
        assert q.ndim == 2 and q.shape[1] == self.dim
        nq = q.shape[0]
        k = int(k)
        idx = np.full((nq, k), -1, dtype=np.int64)
        sims = np.full((nq, k), np.nan, dtype=np.float64)
        cnt = np.zeros(nq, dtype=np.int32)
        lim_p = None
        if row_limit is not None:
            lim = np.ascontiguousarray(row_limit, dtype=np.int64)
            assert lim.shape == (nq,)
            lim_p = lim.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.cslam_bank_search_host(
            self._bank, q.ctypes.data_as(C.c_void_p), dt, nq, k, lim_p, int(mode),
            idx.ctypes.data_as(C.c_void_p), sims.ctypes.data_as(C.c_void_p),
            cnt.ctypes.data_as(C.c_void_p)))
        return idx, sims, cnt

    def add_items_device(self, vectors, items=None):
        """Append float32 descriptors that already live in HBM (torch tensor [m, >=dim])."""
        import torch
        assert vectors.is_cuda and vectors.dtype == torch.float32 and vectors.dim() == 2
        assert vectors.stride(1) == 1
        m = vectors.shape[0]
        if self._bank is None:
            self.device = vectors.device.index or 0
            self._create(vectors.shape[1])
        st = torch.cuda.current_stream(vectors.device).cuda_stream
        _lib.check(self._lib.cslam_bank_add_dev(self._bank, C.c_void_p(vectors.data_ptr()),
                                                vectors.stride(0), m, C.c_void_p(st)))
        items = range(self.n, self.n + m) if items is None else list(items)
        for it in items:
            self.items[self.n] = it
            self.n += 1

    def search_device(self, queries, k, row_limit=None, mode=MODE_AUTO, out=None):
        """Device-resident search: queries torch [nq, dim] float32/float64 on the bank's GPU.
        Returns torch tensors (rows int64 [nq,k], sims float64 [nq,k], cnt int32 [nq])."""
        import torch
        assert queries.is_cuda and queries.dim() == 2 and queries.stride(1) == 1
        dt = _lib.F32 if queries.dtype == torch.float32 else _lib.F64
        assert queries.dtype in (torch.float32, torch.float64)
        nq = queries.shape[0]
        if out is None:
            out = (torch.empty((nq, k), dtype=torch.int64, device=queries.device),
                   torch.empty((nq, k), dtype=torch.float64, device=queries.device),
                   torch.empty((nq,), dtype=torch.int32, device=queries.device))
        lim_p = None
        if row_limit is not None:
            assert row_limit.is_cuda and row_limit.dtype == torch.int64 and row_limit.is_contiguous()
            lim_p = C.c_void_p(row_limit.data_ptr())
        st = torch.cuda.current_stream(queries.device).cuda_stream
        _lib.check(self._lib.cslam_bank_search_dev(
            self._bank, C.c_void_p(queries.data_ptr()), dt, queries.stride(0), nq, int(k), lim_p,
            int(mode), C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()),
            C.c_void_p(out[2].data_ptr()), C.c_void_p(st)))
        return out

    def search_device_async(self, queries, k, row_limit=None, mode=MODE_AUTO, out=None):
        """`search_device` in two halves (C ABI: cslam_bank_search_enqueue_dev / cslam_bank_search_finish): every kernel of
        the search is enqueued on the current stream and the call returns WITHOUT a host synchronisation; the returned
        handle's `finish()` waits only for the event behind the uncertified-query count (work enqueued since -- the next
        step's extraction -- keeps running), enqueues the exact-scan fallback when that count is not zero, and returns the
        same device tensors `search_device` returns (valid in stream order).  One search per bank may be in flight; the
        bank must not change until `finish()`."""
        import torch
        assert queries.is_cuda and queries.dim() == 2 and queries.stride(1) == 1
        assert queries.dtype in (torch.float32, torch.float64)
        dt = _lib.F32 if queries.dtype == torch.float32 else _lib.F64
        nq = queries.shape[0]
        if out is None:
            out = (torch.empty((nq, k), dtype=torch.int64, device=queries.device),
                   torch.empty((nq, k), dtype=torch.float64, device=queries.device),
                   torch.empty((nq,), dtype=torch.int32, device=queries.device))
        lim_p = None
        if row_limit is not None:
            assert row_limit.is_cuda and row_limit.dtype == torch.int64 and row_limit.is_contiguous()
            lim_p = C.c_void_p(row_limit.data_ptr())
        st = torch.cuda.current_stream(queries.device).cuda_stream
        _lib.check(self._lib.cslam_bank_search_enqueue_dev(
            self._bank, C.c_void_p(queries.data_ptr()), dt, queries.stride(0), nq, int(k), lim_p,
            int(mode), C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()),
            C.c_void_p(out[2].data_ptr()), C.c_void_p(st)))
        return PendingSearch([self], out, keep=(queries, row_limit))

    def item_array(self):
        """items of rows [0, n) as an int64 array when every item is an int (keyframe ids), else None: lets the
        batched callers map result rows to keyframe ids with one gather instead of a dict lookup per row."""
        have, arr = self._item_ids
        if arr is None:
            return None
        if have < self.n:
            fresh = [self.items[r] for r in range(have, self.n)]
            if not all(isinstance(x, (int, np.integer)) for x in fresh):
                self._item_ids = (0, None)           # some item is not a keyframe number: dict lookups from now on
                return None
            arr = np.concatenate((arr[:have], np.asarray(fresh, dtype=np.int64)))
            self._item_ids = (self.n, arr)
        return arr

    def last_stats(self):
        """(uncertified queries re-done by the scan, mode used, bank segments, query tiles)."""
        s = (C.c_int64 * 4)()
        _lib.check(self._lib.cslam_bank_last_stats(self._bank, C.byref(s)))
        return tuple(int(x) for x in s)

    def last_stage(self):
        """(fp16 products per pair of the last MFMA-mode search's candidate stage -- 0: the f32-input stage --, searches left on the
        f32-input stage, length of the next back-off, whether CSLAM_MFMA_STAGE1 fixed the stage): cslam_bank_last_stage."""
        s = (C.c_int32 * 4)()
        _lib.check(self._lib.cslam_bank_last_stage(self._bank, C.byref(s)))
        return int(s[0]), int(s[1]), int(s[2]), bool(s[3])

    def last_kernel_ms(self):
        ms = C.c_float(-1.0)
        _lib.check(self._lib.cslam_bank_last_kernel_ms(self._bank, C.byref(ms)))
        return float(ms.value)


class PendingSearch(object):
    """An enqueued search (one bank, or a list of banks sharing one query batch).  `finish()` -> the device output tensors;
    `uncertified` afterwards = queries the exact scan had to re-do (summed over the banks).  Holds the queries / row limits
    alive until then."""

    def __init__(self, banks, out, keep=None, host_copy=None):
        self._banks, self._out, self._keep, self._host_copy = banks, out, keep, host_copy
        self.uncertified = None

    def finish(self):
        if self._banks is None:
            return self._out
        banks, self._banks = self._banks, None
        lib = banks[0]._lib
        n = C.c_int64(0)
        if len(banks) == 1:
            _lib.check(lib.cslam_bank_search_finish(banks[0]._bank, C.byref(n)))
        else:
            handles = (C.c_void_p * len(banks))(*[b._bank for b in banks])
            _lib.check(lib.cslam_bank_search_multi_finish(handles, len(banks), C.byref(n)))
        self.uncertified = int(n.value)
        self._keep = None
        return self._out

    @property
    def out(self):
        """The output tensors as enqueued: valid in stream order, PROVISIONAL for queries the certificate could not settle
        until `finish()` has re-done them."""
        return self._out

    def uncertified_to(self, count):
        """Enqueue (current stream) a copy of this search's uncertified-query count into `count`, an int32 CUDA tensor of one
        element: the device-side twin of `finish()`'s return, for a caller that ships it with the provisional lists."""
        import torch
        assert self._banks is not None and len(self._banks) == 1, "one pending single-bank search"
        assert count.is_cuda and count.dtype == torch.int32 and count.numel() == 1
        b = self._banks[0]
        st = torch.cuda.current_stream(count.device).cuda_stream
        _lib.check(b._lib.cslam_bank_search_flag_copy_dev(b._bank, C.c_void_p(count.data_ptr()), C.c_void_p(st)))

    def __del__(self):                       # a dropped handle must not leave the bank locked
        try:
            self.finish()
        except Exception:
            pass


def search_multi_device(banks, queries, ks, row_limits=None, mode=MODE_AUTO, defer=False):
    """One batch of device-resident queries against several banks of the same GPU in ONE library call
    (`cslam_bank_search_multi_enqueue_dev` + `_finish`): the kernels of every bank are enqueued before the host waits for
    the uncertified-query counts, and the results come back in three copies.  defer=True returns a handle right after the
    enqueue; its `finish()` does the rest (a batch host runs the next step's extraction in between).  banks: NearestNeighborsMatching objects (populated, same dim and
    device); ks: k per bank; row_limits: per bank None or an int64 CUDA tensor [nq].
    Returns a list of (rows [nq,k] int64, sims [nq,k] float64, cnt [nq] int32) NUMPY arrays, one per bank -- what
    `search_device` of each bank would give, downloaded."""
    import torch
    assert queries.is_cuda and queries.dim() == 2 and queries.stride(1) == 1
    assert queries.dtype in (torch.float32, torch.float64)
    nb, nq = len(banks), queries.shape[0]
    assert nb >= 1 and len(ks) == nb and all(b._bank is not None for b in banks)
    lib = banks[0]._lib
    dt = _lib.F32 if queries.dtype == torch.float32 else _lib.F64
    ks = [int(k) for k in ks]
    offs = np.concatenate(([0], np.cumsum([nq * k for k in ks]))).tolist()
    dev = queries.device
    idx = torch.empty(offs[-1], dtype=torch.int64, device=dev)
    sims = torch.empty(offs[-1], dtype=torch.float64, device=dev)
    cnt = torch.empty(nb * nq, dtype=torch.int32, device=dev)
    handles = (C.c_void_p * nb)(*[b._bank for b in banks])
    kk = (C.c_int * nb)(*ks)
    lims = None
    if row_limits is not None and any(r is not None for r in row_limits):
        for r in row_limits:
            assert r is None or (r.is_cuda and r.dtype == torch.int64 and r.is_contiguous() and r.shape == (nq,))
        lims = (C.c_void_p * nb)(*[None if r is None else r.data_ptr() for r in row_limits])
    p_idx = (C.c_void_p * nb)(*[idx.data_ptr() + 8 * offs[i] for i in range(nb)])
    p_sim = (C.c_void_p * nb)(*[sims.data_ptr() + 8 * offs[i] for i in range(nb)])
    p_cnt = (C.c_void_p * nb)(*[cnt.data_ptr() + 4 * nq * i for i in range(nb)])
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.cslam_bank_search_multi_enqueue_dev(handles, nb, C.c_void_p(queries.data_ptr()), dt, queries.stride(0), nq,
                                                       kk, lims, int(mode), p_idx, p_sim, p_cnt, C.c_void_p(st)))
    pend = PendingSearch(list(banks), (idx, sims, cnt), keep=(queries, row_limits))
    if not defer:
        return PendingMultiSearch(pend, offs, nq, ks).finish()
    # deferred: the result copies are put on the stream NOW, behind the search and ahead of whatever the caller enqueues
    # next, into pinned memory; finish() then waits for this event only.  (They are final unless a query failed its
    # certificate -- finish() re-copies in that case.)
    owned = []
    host = tuple(_pinned(t.shape, t.dtype, owned) for t in (idx, sims, cnt))
    for h, t in zip(host, (idx, sims, cnt)):
        h.copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    return PendingMultiSearch(pend, offs, nq, ks, host=host, event=ev, owned=owned)


_PINNED_POOL = []       # free pinned staging buffers (uint8); a live deferred search owns the ones it took


def _pinned_take(nbytes):
    """A pinned host buffer of at least nbytes from the pool (hipHostMalloc is too slow to sit on the per-chunk path); the
    handle that takes it gives it back in finish(), so several deferred searches may be outstanding at once."""
    import torch
    best = None
    for i, b in enumerate(_PINNED_POOL):
        if b.numel() >= nbytes and (best is None or b.numel() < _PINNED_POOL[best].numel()):
            best = i
    if best is not None:
        return _PINNED_POOL.pop(best)
    return torch.empty(max(int(nbytes * 1.25), 4096), dtype=torch.uint8, pin_memory=True)


def _pinned(shape, dtype, owner):
    import torch
    n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
    buf = _pinned_take(n)
    owner.append(buf)
    return buf[:n].view(dtype).view(shape)


class PendingMultiSearch(object):
    """`search_multi_device(..., defer=True)`: `finish()` -> the list of per-bank NUMPY results."""

    def __init__(self, pend, offs, nq, ks, host=None, event=None, owned=None):
        self._pend, self._offs, self._nq, self._ks = pend, offs, nq, ks
        self._host, self._event, self._owned = host, event, owned
        self._result = None

    def finish(self):
        if self._result is None:
            idx, sims, cnt = self._pend.finish()
            if self._host is not None:
                self._event.synchronize()                      # the copies issued right behind the search: nothing later
            if self._host is not None and self._pend.uncertified == 0:
                h_idx, h_sims, h_cnt = (h.numpy().copy() for h in self._host)
            else:                                              # a fallback rewrote some rows after the early copies
                h_idx, h_sims, h_cnt = idx.cpu().numpy(), sims.cpu().numpy(), cnt.cpu().numpy()
            if self._owned:
                _PINNED_POOL.extend(self._owned)               # copies are done (event waited for): back to the pool
                self._owned = self._host = None
            offs, nq, ks = self._offs, self._nq, self._ks
            self._result = [(h_idx[offs[i]:offs[i + 1]].reshape(nq, ks[i]), h_sims[offs[i]:offs[i + 1]].reshape(nq, ks[i]),
                             h_cnt[i * nq:(i + 1) * nq]) for i in range(len(ks))]
            self._pend = None
        return self._result
