"""Multi-GPU matching over RCCL, two shapes:
`ShardedInterRobotMatcher` -- one robot bank per GPU (BASELINE config 4), described below;
`RowShardedBankMatcher`    -- ONE bank split by rows over the GPUs (the single-bank metric at > 1 GPU), further down.

One-robot-bank-per-GPU inter-robot matching:

Rank g plays robot g: it owns bank B_g in its HBM and produces its own new descriptors.
Per step every rank contributes its new descriptors Q_g; ONE all-gather (RCCL over xGMI,
torch.distributed backend "nccl") gives every rank all queries; rank g then scores
    its own rows      -> top-k   against B_g   (intra-robot, lcsm.py:74-92)
    every other robot -> best-1  against B_g   (lcsm.py:56-72: remote descriptor vs local bank)
in one launch over all N*m queries (the best-1 is the head of the top-k list).
The reference needs best-1 per (query, bank) pair only, so results stay on the rank that
owns the bank: no second collective, no reduction (the reference exchanges descriptors over
ROS 2 topics, gdlcd.py:198-227,407-422; this replaces that transport inside one node).
`search_fn` / `gather_fn` are injectable so the control flow is testable on CPU with gloo.
"""
import torch
import torch.distributed as dist


def all_gather_rows(local, world_size, group=None):
    """[m, d] on every rank -> [world_size*m, d] (rank-major).  Equal m on all ranks."""
    if world_size == 1:
        return local
    out = torch.empty((world_size * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


class _Done(object):
    """Handle of a collective that has already completed (injected synchronous functions, world size 1)."""

    def wait(self):
        return True


def all_gather_rows_async(local, world_size, group=None):
    """all_gather_rows issued without waiting: returns (out, handle).  On RCCL the collective runs on the process
    group's own stream, ordered after what the current stream has enqueued so far; `handle.wait()` makes the current
    stream wait for it -- kernels launched between the two overlap with the transfer."""
    if world_size == 1:
        return local, _Done()
    out = torch.empty((world_size * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
    return out, dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=True)


def exchange_lists_async(packed, world_size, group=None):
    """exchange_lists (below) issued without waiting: (out, handle)."""
    if world_size == 1:
        return packed, _Done()
    out = torch.empty_like(packed)
    return out, dist.all_to_all_single(out, packed.contiguous(), group=group, async_op=True)


class PendingStep(object):
    """A sharded step whose kernels and collectives are enqueued (`step_begin`); `finish()` -> what `step()` returns.
    Between the two the host is free -- it enqueues the NEXT step's extraction, so the GPU queue never drains on a host
    wait (the N = 1 step's `search_device_async` pipelining carried over to N > 1)."""

    def __init__(self, finish_fn):
        self._finish, self._result = finish_fn, None

    def finish(self):
        if self._finish is not None:
            self._result = self._finish()
            self._finish = None
        return self._result


class _SyncPending(object):
    """A search that has already run (a synchronous `search_fn` behind the asynchronous interface)."""

    def __init__(self, out):
        self.out, self.uncertified = out, 0

    def uncertified_to(self, count):
        count.zero_()

    def finish(self):
        return self.out


def _chunk_bounds(m, chunks):
    """[0, m) cut into at most `chunks` equal pieces (the same cut on every rank: m is equal on all ranks)."""
    c = max(1, min(int(chunks), m)) if m > 0 else 1
    step = -(-m // c) if m > 0 else 0
    return [(a, min(a + step, m)) for a in range(0, m, step)] if m > 0 else [(0, 0)]


class ShardedInterRobotMatcher(object):
    def __init__(self, rank, world_size, search_fn, k_intra=5, gather_fn=None, chunks=2, search_async_fn=None):
        """search_fn(queries [nq,d], k) -> (rows [nq,k], sims [nq,k], cnt [nq]) against THIS
        rank's bank (e.g. NearestNeighborsMatching.search_device).
        search_async_fn(queries, k) -> handle with `.finish()` -> the same triple (NearestNeighborsMatching.
        search_device_async): what `step_begin` enqueues; None = `search_fn` behind the same interface.
        chunks: the step's descriptors are exchanged and scored in this many pieces, the all-gather of piece
        t+1 in flight while piece t is scored (SURVEY 8e "double-buffer query chunks"); 1 = one gather, one launch.
        gather_fn: injected synchronous all-gather (tests, host-staged debug runs); None = RCCL, asynchronous."""
        self.rank, self.world = rank, world_size
        self.search_fn, self.gather_fn, self.k_intra = search_fn, gather_fn, k_intra
        self.search_async_fn = search_async_fn
        self.chunks = int(chunks) if world_size > 1 else 1

    def _gather(self, x):
        if self.gather_fn is not None:
            return self.gather_fn(x, self.world), _Done()
        return all_gather_rows_async(x, self.world)

    def _search_async(self, q, k):
        if self.search_async_fn is not None:
            return self.search_async_fn(q, k)
        return _SyncPending(self.search_fn(q, k))

    def _split(self, rows, sims, cnt, m):
        """[world*m, k] robot-major results -> (intra, inter) as `step` documents."""
        k = rows.shape[1]
        rows, sims, cnt = rows.view(self.world, m, k), sims.view(self.world, m, k), cnt.view(self.world, m)
        intra = (rows[self.rank], sims[self.rank], cnt[self.rank])
        keep = [g for g in range(self.world) if g != self.rank]
        robot = torch.tensor(keep, device=rows.device).repeat_interleave(m)
        inter = (rows[keep].reshape(-1, k)[:, :1], sims[keep].reshape(-1, k)[:, :1],
                 cnt[keep].reshape(-1).clamp(max=1), robot)
        return intra, inter

    def step_begin(self, local_desc):
        """`step` in two halves: the all-gather and ONE search over every robot's descriptors are enqueued here (no host
        wait: the bank takes one search at a time, so the step is one piece and the gather is not hidden under a previous
        piece's search -- 17 MB per rank against the extraction the caller enqueues next); `finish()` of the returned
        PendingStep waits for the search's certificate count only (cslam_bank_search_finish) and returns `step`'s result.
        Results stay with the bank owner: no second collective, so nothing provisional ever leaves the rank."""
        m = local_desc.shape[0]
        if self.world == 1:
            pend = self._search_async(local_desc, self.k_intra)
            return PendingStep(lambda: (pend.finish(), None))
        if m == 0:
            empty = self.step(local_desc)
            return PendingStep(lambda: empty)
        allq, handle = self._gather(local_desc)
        handle.wait()
        pend = self._search_async(allq, self.k_intra)
        return PendingStep(lambda: self._split(*pend.finish(), m))

    def step(self, local_desc):
        """local_desc [m, d]: this robot's new descriptors.  Returns
        (intra (rows, sims, cnt) for the m local queries,
         inter (rows, sims, cnt, robot_of_query [nq_remote]) for all other robots' queries, robot-major)."""
        m = local_desc.shape[0]
        if self.world == 1:
            return self.search_fn(local_desc, self.k_intra), None
        if m == 0:                                                   # an empty step (m is equal on all ranks): no collective
            dev, k = local_desc.device, self.k_intra
            e = lambda dt, *shape: torch.empty(shape, dtype=dt, device=dev)
            return ((e(torch.int64, 0, k), e(torch.float64, 0, k), e(torch.int32, 0)),
                    (e(torch.int64, 0, 1), e(torch.float64, 0, 1), e(torch.int32, 0), e(torch.int64, 0)))
        bounds = _chunk_bounds(m, self.chunks)
        inflight = self._gather(local_desc[bounds[0][0]:bounds[0][1]])
        parts = []
        for t, (a, b) in enumerate(bounds):
            allq, handle = inflight
            if t + 1 < len(bounds):                                  # next piece on the wire before this one is scored
                inflight = self._gather(local_desc[bounds[t + 1][0]:bounds[t + 1][1]])
            handle.wait()
            # ONE launch over every robot's piece: the best-1 the remote rows need is the head of the same top-k
            # list the local rows need in full (same scores, same order), so nothing is computed twice
            rows, sims, cnt = self.search_fn(allq, self.k_intra)
            c = b - a
            # explicit sizes: view(..., -1) cannot infer a dimension of an empty step (m == 0 is equal on all ranks)
            parts.append((rows.view(self.world, c, rows.shape[1]), sims.view(self.world, c, sims.shape[1]), cnt.view(self.world, c)))
        rows = torch.cat([p[0] for p in parts], dim=1)               # [world, m, k], robot-major like one big gather
        sims = torch.cat([p[1] for p in parts], dim=1)
        cnt = torch.cat([p[2] for p in parts], dim=1)
        intra = (rows[self.rank], sims[self.rank], cnt[self.rank])
        keep = [g for g in range(self.world) if g != self.rank]
        robot = torch.tensor(keep, device=rows.device).repeat_interleave(m)
        k = rows.shape[2]
        inter = (rows[keep].reshape(-1, k)[:, :1], sims[keep].reshape(-1, k)[:, :1],
                 cnt[keep].reshape(-1).clamp(max=1), robot)
        return intra, inter


# ---------------------------------------------------------------------------------------------------------
# ONE bank row-sharded over the ranks ("for the single-bank 100k metric at > 1 GPU: row-shard
# the bank, each GPU computes local top-k, then merge")
# ---------------------------------------------------------------------------------------------------------
def exchange_lists(packed, world_size, group=None):
    """packed [world_size, m, w] int64 on every rank: slice r = this shard's lists for rank r's queries.
    Returns [world_size, m, w]: slice s = shard s's lists for THIS rank's queries.  One all-to-all (RCCL),
    m*w*8 bytes per pair of ranks -- the only traffic besides the all-gather of the descriptors."""
    if world_size == 1:
        return packed
    out = torch.empty_like(packed)
    dist.all_to_all_single(out, packed.contiguous(), group=group)
    return out


def merge_topk_device(rows, sims, cnt, row_offsets):
    """rows / sims [S, m, k], cnt [S, m] (device tensors, shard-local rows) -> (rows [m,k] GLOBAL, sims [m,k],
    cnt [m]) through `cslam_topk_merge_dev` (csrc/bank.hip); there is no CPU path."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    S, m, k = rows.shape
    assert rows.is_cuda and rows.dtype == torch.int64 and sims.dtype == torch.float64 and cnt.dtype == torch.int32
    rows, sims, cnt = rows.contiguous(), sims.contiguous(), cnt.contiguous()
    out = (torch.empty((m, k), dtype=torch.int64, device=rows.device),
           torch.empty((m, k), dtype=torch.float64, device=rows.device),
           torch.empty((m,), dtype=torch.int32, device=rows.device))
    off = (C.c_int64 * S)(*[int(o) for o in row_offsets])
    st = torch.cuda.current_stream(rows.device).cuda_stream
    _lib.check(lib.cslam_topk_merge_dev(C.c_void_p(rows.data_ptr()), C.c_void_p(sims.data_ptr()),
                                        C.c_void_p(cnt.data_ptr()), off, S, m, k, C.c_void_p(out[0].data_ptr()),
                                        C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()), C.c_void_p(st)))
    return out


class RowShardedBankMatcher(object):
    """One descriptor bank whose rows are split over the ranks: rank g holds the global rows
    [row_offsets[g], row_offsets[g+1]).  Per step every rank contributes m new descriptors and receives their
    exact top-k over the WHOLE bank -- the result `NearestNeighborsMatching.search` (nns_matching.py:42-61)
    gives on the unsharded bank, indices and float64 scores bit for bit:
        1. all-gather of the descriptors (as ShardedInterRobotMatcher);
        2. every rank: top-k of all world*m queries against its shard, one launch (pair work per rank =
           world*m x n/world, independent of the rank count);
        3. all-to-all of the (rows, scores, count) lists, world*m*(2k+1)*8 bytes per rank;
        4. `cslam_topk_merge_dev`: k best of the world*k contenders per query, global row numbers.
    `search_fn`, `gather_fn`, `exchange_fn`, `merge_fn` are injectable (CPU control-flow tests with gloo)."""

    def __init__(self, rank, world_size, search_fn, row_offsets, k=5, gather_fn=None, exchange_fn=None,
                 merge_fn=merge_topk_device, chunks=2, search_async_fn=None):
        """chunks: pieces per step.  The all-gather of piece t+1 and the all-to-all of piece t-1's lists are in
        flight while piece t is scored; both collectives of a piece are issued in the same order on every rank.
        gather_fn / exchange_fn: injected synchronous collectives (tests, host-staged debug runs); None = RCCL,
        asynchronous."""
        assert len(row_offsets) >= world_size
        self.rank, self.world, self.k = rank, world_size, int(k)
        self.row_offsets = [int(o) for o in row_offsets[:world_size]]
        self.search_fn, self.gather_fn, self.exchange_fn, self.merge_fn = search_fn, gather_fn, exchange_fn, merge_fn
        self.search_async_fn = search_async_fn
        self.chunks = int(chunks) if world_size > 1 else 1

    def _gather(self, x):
        if self.gather_fn is not None:
            return self.gather_fn(x, self.world), _Done()
        return all_gather_rows_async(x, self.world)

    def _exchange(self, packed):
        if self.exchange_fn is not None:
            return self.exchange_fn(packed, self.world), _Done()
        return exchange_lists_async(packed, self.world)

    def _search_async(self, q, k):
        if self.search_async_fn is not None:
            return self.search_async_fn(q, k)
        return _SyncPending(self.search_fn(q, k))

    @staticmethod
    def _pack(rows, sims, cnt):
        """one int64 buffer per query: k rows | k score bit patterns | count"""
        return torch.cat((rows, sims.contiguous().view(torch.int64), cnt.to(torch.int64)[:, None]), dim=1)

    def step_begin(self, local_desc):
        """`step` in two halves (ONE piece: the bank takes one search at a time).  Enqueued here, without a host wait:
        all-gather, the shard's search, the all-to-all of its lists, the merge.  The lists are PROVISIONAL for queries the
        candidate stage's certificate could not settle (normally none); the shard's count of them travels with the lists as
        one extra row, so after the all-to-all every rank holds every shard's count -- in stream order, and the same numbers
        on every rank.  `finish()`: one event wait for those counts (behind the merge), the banks' own `finish`; only if some
        shard's count is not zero, ALL ranks redo exchange and merge with the re-scanned lists (a collective every rank
        enters, because every rank saw the same counts)."""
        m, k, G = local_desc.shape[0], self.k, self.world
        if G == 1:
            pend = self._search_async(local_desc, k)

            def fin1():
                rows, sims, cnt = pend.finish()
                return self.merge_fn(rows[None], sims[None], cnt[None], self.row_offsets) if self.row_offsets[0] else \
                    (rows, sims, cnt)
            return PendingStep(fin1)
        if m == 0:
            empty = self.step(local_desc)
            return PendingStep(lambda: empty)
        dev = local_desc.device
        allq, handle = self._gather(local_desc)
        handle.wait()
        pend = self._search_async(allq, k)
        flag = torch.zeros((1,), dtype=torch.int32, device=dev)
        pend.uncertified_to(flag)
        w = 2 * k + 1
        packed = torch.empty((G, m + 1, w), dtype=torch.int64, device=dev)
        packed[:, :m] = self._pack(*pend.out).view(G, m, w)
        packed[:, m] = flag.to(torch.int64)                              # the same count to every destination rank
        got, h2 = self._exchange(packed)
        merged = self._merge_piece(got[:, :m], h2)
        owed = got[:, m, 0].max().reshape(1)                             # some shard still owes a re-scan?
        host = torch.empty((1,), dtype=torch.int64, pin_memory=True) if dev.type == "cuda" else torch.empty((1,), dtype=torch.int64)
        host.copy_(owed, non_blocking=True)
        ev = st = None
        if dev.type == "cuda":
            st = torch.cuda.current_stream(dev)                          # the stream everything above was enqueued on
            ev = torch.cuda.Event()
            ev.record(st)

        def redo():
            rows, sims, cnt = pend.finish()                              # unlocks the bank; re-scans what it flagged
            if int(host.item()) == 0:
                return merged
            packed2 = self._pack(rows, sims, cnt).view(G, m, w)
            got2, h3 = self._exchange(packed2.contiguous())
            return self._merge_piece(got2, h3)

        def fin():
            if ev is None:
                return redo()
            ev.synchronize()
            # The bank's re-scan runs on the stream the search was enqueued on; the second exchange and merge read its lists.
            # Run them on THAT stream whatever stream the caller finishes on, and order the caller's stream behind them.
            cur = torch.cuda.current_stream(dev)
            with torch.cuda.stream(st):
                out = redo()
                done = torch.cuda.Event()
                done.record(st)
            if cur != st:
                cur.wait_event(done)
                # the result was allocated from `st`'s pool and will be read on `cur`: tell the caching allocator, or a block freed
                # by the caller could be handed to later work on `st` while kernels on `cur` still read it (round-5 advisor)
                for t_ in (out if isinstance(out, (tuple, list)) else (out,)):
                    if torch.is_tensor(t_) and t_.is_cuda:
                        t_.record_stream(cur)
            return out
        return PendingStep(fin)

    def _merge_piece(self, got, handle):
        handle.wait()
        k = self.k
        return self.merge_fn(got[:, :, :k].contiguous(), got[:, :, k:2 * k].contiguous().view(torch.float64),
                             got[:, :, 2 * k].to(torch.int32).contiguous(), self.row_offsets)

    def step(self, local_desc):
        """local_desc [m, d] -> (rows [m,k] int64 global (-1 padded), sims [m,k] float64, cnt [m] int32)."""
        m, k, G = local_desc.shape[0], self.k, self.world
        if G == 1:
            rows, sims, cnt = self.search_fn(local_desc, k)
            return self.merge_fn(rows[None], sims[None], cnt[None], self.row_offsets) if self.row_offsets[0] else \
                (rows, sims, cnt)
        if m == 0:                                                       # an empty step (m is equal on all ranks): no collective
            dev = local_desc.device
            return (torch.empty((0, k), dtype=torch.int64, device=dev), torch.empty((0, k), dtype=torch.float64, device=dev),
                    torch.empty((0,), dtype=torch.int32, device=dev))
        bounds = _chunk_bounds(m, self.chunks)
        inflight = self._gather(local_desc[bounds[0][0]:bounds[0][1]])
        lists, out = None, []
        for t, (a, b) in enumerate(bounds):
            allq, handle = inflight
            if t + 1 < len(bounds):
                inflight = self._gather(local_desc[bounds[t + 1][0]:bounds[t + 1][1]])
            handle.wait()
            rows, sims, cnt = self.search_fn(allq, k)                    # shard-local rows, [G*(b-a), k]
            if lists is not None:                                        # previous piece's lists arrived meanwhile
                out.append(self._merge_piece(*lists))
            packed = self._pack(rows, sims, cnt)                         # ONE collective per piece
            lists = self._exchange(packed.view(G, b - a, 2 * k + 1))     # -> [G shards, own queries of the piece, 2k+1]
        out.append(self._merge_piece(*lists))
        if len(out) == 1:
            return out[0]
        return tuple(torch.cat([o[i] for o in out]) for i in range(3))
