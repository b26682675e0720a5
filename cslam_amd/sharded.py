"""One-robot-bank-per-GPU inter-robot matching over RCCL (BASELINE config 4).

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
