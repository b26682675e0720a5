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


class ShardedInterRobotMatcher(object):
    def __init__(self, rank, world_size, search_fn, k_intra=5, gather_fn=all_gather_rows):
        """search_fn(queries [nq,d], k) -> (rows [nq,k], sims [nq,k], cnt [nq]) against THIS
        rank's bank (e.g. NearestNeighborsMatching.search_device)."""
        self.rank, self.world = rank, world_size
        self.search_fn, self.gather_fn, self.k_intra = search_fn, gather_fn, k_intra

    def step(self, local_desc):
        """local_desc [m, d]: this robot's new descriptors.  Returns
        (intra (rows, sims, cnt) for the m local queries,
         inter (rows, sims, cnt, robot_of_query [nq_remote]) for all other robots' queries)."""
        m = local_desc.shape[0]
        allq = self.gather_fn(local_desc, self.world)
        if self.world == 1:
            return self.search_fn(allq, self.k_intra), None
        # ONE launch over every robot's new descriptors: the best-1 the remote rows need is the head of the
        # same top-k list the local rows need in full (same scores, same order), so nothing is computed twice
        rows, sims, cnt = self.search_fn(allq, self.k_intra)
        lo, hi = self.rank * m, (self.rank + 1) * m
        intra = (rows[lo:hi], sims[lo:hi], cnt[lo:hi])
        robot = torch.arange(self.world, device=allq.device).repeat_interleave(m)
        cut = lambda t: torch.cat((t[:lo], t[hi:]))             # noqa: E731 -- everyone's rows except this robot's
        inter = (cut(rows)[:, :1], cut(sims)[:, :1], cut(cnt).clamp(max=1), cut(robot))
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

    def __init__(self, rank, world_size, search_fn, row_offsets, k=5, gather_fn=all_gather_rows,
                 exchange_fn=exchange_lists, merge_fn=merge_topk_device):
        assert len(row_offsets) >= world_size
        self.rank, self.world, self.k = rank, world_size, int(k)
        self.row_offsets = [int(o) for o in row_offsets[:world_size]]
        self.search_fn, self.gather_fn, self.exchange_fn, self.merge_fn = search_fn, gather_fn, exchange_fn, merge_fn

    def step(self, local_desc):
        """local_desc [m, d] -> (rows [m,k] int64 global (-1 padded), sims [m,k] float64, cnt [m] int32)."""
        m, k, G = local_desc.shape[0], self.k, self.world
        allq = self.gather_fn(local_desc, G)
        rows, sims, cnt = self.search_fn(allq, k)                       # shard-local rows, [G*m, k]
        if G == 1:
            return self.merge_fn(rows[None], sims[None], cnt[None], self.row_offsets) if self.row_offsets[0] else \
                (rows, sims, cnt)
        # one int64 buffer per query: k rows | k score bit patterns | count  -> ONE collective
        packed = torch.cat((rows, sims.contiguous().view(torch.int64), cnt.to(torch.int64)[:, None]), dim=1)
        got = self.exchange_fn(packed.view(G, m, 2 * k + 1), G)          # [G shards, m own queries, 2k+1]
        return self.merge_fn(got[:, :, :k].contiguous(), got[:, :, k:2 * k].contiguous().view(torch.float64),
                             got[:, :, 2 * k].to(torch.int32).contiguous(), self.row_offsets)
