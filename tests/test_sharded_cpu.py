"""CPU suite: the N > 1 control flow (one robot bank per rank, all-gather of new descriptors,
local top-k / best-1) on 2 gloo ranks, with the CPU oracle injected as the search function."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, outdir, chunks=2):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cslam_amd.sharded import ShardedInterRobotMatcher
    from oracle import pyoracle
    from helpers import unit_rows
    bank = unit_rows(np.random.default_rng(1234 + rank), 500, 64)

    def search(q, k):
        i, s, c = pyoracle.nns_search(bank, q.numpy(), k)
        return torch.from_numpy(i), torch.from_numpy(s), torch.from_numpy(c)

    m = ShardedInterRobotMatcher(rank, world, search, k_intra=5, chunks=chunks)
    local = torch.from_numpy(unit_rows(np.random.default_rng(4321 + rank), 40, 64))
    intra, inter = m.step(local)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), intra_rows=intra[0].numpy(), intra_sims=intra[1].numpy(),
             inter_rows=inter[0].numpy(), inter_sims=inter[1].numpy(), inter_robot=inter[3].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks", [(2, 1), (2, 2), (4, 3), (8, 2)])
def test_gloo_one_bank_per_rank_step(tmp_path, world, chunks):
    """2 / 4 / 8 ranks (config 4 has 8 robots), one piece or several overlapped pieces per step: every rank's own
    keyframes get their top-5 and every other robot's keyframes their best-1 against this rank's bank, equal to the
    oracle; the asynchronous gloo all-gather is the default path (no injected gather)."""
    from helpers import unit_rows
    from oracle import pyoracle
    port = 29500 + (os.getpid() % 1000) + 7 * world + chunks
    mp.spawn(_worker, args=(world, port, str(tmp_path), chunks), nprocs=world, join=True)
    banks = [unit_rows(np.random.default_rng(1234 + r), 500, 64) for r in range(world)]
    qs = [unit_rows(np.random.default_rng(4321 + r), 40, 64) for r in range(world)]
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npz")
        i, s, _ = pyoracle.nns_search(banks[r], qs[r], 5)           # own keyframes vs own bank
        assert np.array_equal(got["intra_rows"], i) and np.array_equal(got["intra_sims"], s)
        others = [o for o in range(world) if o != r]
        oq = np.concatenate([qs[o] for o in others])                # the other robots' keyframes vs own bank
        i, s, _ = pyoracle.nns_search(banks[r], oq, 1)
        assert np.array_equal(got["inter_rows"], i) and np.array_equal(got["inter_sims"], s)
        assert np.array_equal(got["inter_robot"], np.repeat(others, 40))


def test_single_rank_is_passthrough():
    from cslam_amd.sharded import ShardedInterRobotMatcher
    calls = []

    def search(q, k):
        calls.append((tuple(q.shape), k))
        return q, q, q

    m = ShardedInterRobotMatcher(0, 1, search, k_intra=7)
    intra, inter = m.step(torch.zeros(3, 8))
    assert inter is None and calls == [((3, 8), 7)]


# ---- one bank row-sharded over the ranks (the single-bank metric at > 1 GPU) ----------------
def _host_merge(rows, sims, cnt, row_offsets):
    """Test stand-in for cslam_topk_merge_dev (the product merge is the HIP kernel; tests/test_nns_gpu.py checks
    that one): k best of the shards' lists, descending similarity, NaN first, ties -> larger global row."""
    S, m, k = rows.shape
    out_r = np.full((m, k), -1, np.int64)
    out_s = np.full((m, k), np.nan)
    out_c = np.zeros(m, np.int32)
    for q in range(m):
        ent = [(np.inf if np.isnan(sims[s, q, e]) else sims[s, q, e], int(rows[s, q, e]) + int(row_offsets[s]),
                sims[s, q, e]) for s in range(S) for e in range(int(cnt[s, q]))]
        ent.sort(key=lambda t: (t[0], t[1]), reverse=True)
        for j, (_, r, v) in enumerate(ent[:k]):
            out_r[q, j], out_s[q, j] = r, v
        out_c[q] = min(k, len(ent))
    return torch.from_numpy(out_r), torch.from_numpy(out_s), torch.from_numpy(out_c)


def _row_worker(rank, world, port, outdir, shard_rows, chunks=2):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cslam_amd.sharded import RowShardedBankMatcher
    from oracle import pyoracle
    from helpers import unit_rows
    offs = np.concatenate(([0], np.cumsum(shard_rows)))
    whole = unit_rows(np.random.default_rng(99), int(offs[-1]), 64)
    shard = whole[offs[rank]:offs[rank + 1]]

    def search(q, k):
        i, s, c = pyoracle.nns_search(shard, q.numpy(), k)
        return torch.from_numpy(i), torch.from_numpy(s), torch.from_numpy(c)

    merge = lambda r, s, c, o: _host_merge(r.numpy(), s.numpy(), c.numpy(), o)      # noqa: E731
    m = RowShardedBankMatcher(rank, world, search, offs[:world], k=5, merge_fn=merge, chunks=chunks)
    local = torch.from_numpy(unit_rows(np.random.default_rng(4321 + rank), 40, 64))
    rows, sims, cnt = m.step(local)
    np.savez(os.path.join(outdir, f"s{rank}.npz"), rows=rows.numpy(), sims=sims.numpy(), cnt=cnt.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shard_rows,chunks", [((300, 300), 1), ((597, 3), 2), ((150, 150, 200, 100), 3),
                                               ((75, 75, 75, 75, 75, 75, 147, 3), 2)])
def test_gloo_row_sharded_bank(tmp_path, shard_rows, chunks):
    """2 / 4 / 8 ranks: each rank's keyframes get the top-k of the WHOLE bank: indices and float64 scores identical
    to the oracle on the unsharded bank (the 3-row shard returns fewer than k entries to the merge); one piece or
    several overlapped pieces per step, collectives through the default asynchronous path."""
    from helpers import unit_rows
    from oracle import pyoracle
    world, port = len(shard_rows), 30500 + (os.getpid() % 1000) + 11 * len(shard_rows) + chunks
    mp.spawn(_row_worker, args=(world, port, str(tmp_path), shard_rows, chunks), nprocs=world, join=True)
    whole = unit_rows(np.random.default_rng(99), sum(shard_rows), 64)
    for r in range(world):
        q = unit_rows(np.random.default_rng(4321 + r), 40, 64)
        i, s, c = pyoracle.nns_search(whole, q, 5)
        got = np.load(tmp_path / f"s{r}.npz")
        assert np.array_equal(got["rows"], i) and np.array_equal(got["sims"], s) and np.array_equal(got["cnt"], c)


class _FakePending(object):
    """A pending search as NearestNeighborsMatching.search_device_async hands out: `.out` is valid in stream order but
    PROVISIONAL for `n_bad` queries (here: deliberately wrong rows and scores) until `.finish()` has re-done them."""

    def __init__(self, exact, n_bad):
        self._exact, self.n_bad, self.finished = exact, n_bad, False
        rows, sims, cnt = (t.clone() for t in exact)
        rows[:n_bad] = 0
        sims[:n_bad] = 0.999                   # would win every merge if it were believed
        self.out = (rows, sims, cnt)

    def uncertified_to(self, count):
        count.fill_(self.n_bad)

    def finish(self):
        self.finished = True
        return self._exact


def _row_worker_async(rank, world, port, outdir, shard_rows, bad_rank):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cslam_amd.sharded import RowShardedBankMatcher, ShardedInterRobotMatcher
    from oracle import pyoracle
    from helpers import unit_rows
    offs = np.concatenate(([0], np.cumsum(shard_rows)))
    whole = unit_rows(np.random.default_rng(99), int(offs[-1]), 64)
    shard = whole[offs[rank]:offs[rank + 1]]
    pends = []

    def search_async(q, k):
        i, s, c = pyoracle.nns_search(shard, q.numpy(), k)
        pends.append(_FakePending((torch.from_numpy(i), torch.from_numpy(s), torch.from_numpy(c)), 7 if rank == bad_rank else 0))
        return pends[-1]

    merge = lambda r, s, c, o: _host_merge(r.numpy(), s.numpy(), c.numpy(), o)      # noqa: E731
    m = RowShardedBankMatcher(rank, world, None, offs[:world], k=5, merge_fn=merge, search_async_fn=search_async)
    local = [torch.from_numpy(unit_rows(np.random.default_rng(4321 + 10 * t + rank), 40, 64)) for t in range(2)]
    h0 = m.step_begin(local[0])
    h1 = m.step_begin(local[1])                 # the next step is enqueued before the first is finished
    out = [h0.finish(), h1.finish()]
    assert all(p.finished for p in pends)
    # one robot bank per rank through the same two halves
    r = ShardedInterRobotMatcher(rank, world, None, k_intra=5, search_async_fn=search_async)
    intra, inter = r.step_begin(local[0]).finish()
    np.savez(os.path.join(outdir, f"a{rank}.npz"), rows0=out[0][0].numpy(), sims0=out[0][1].numpy(), cnt0=out[0][2].numpy(),
             rows1=out[1][0].numpy(), sims1=out[1][1].numpy(), intra_rows=intra[0].numpy(), inter_rows=inter[0].numpy(),
             inter_sims=inter[1].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shard_rows,bad_rank", [((300, 300), -1), ((150, 150, 200, 100), 2)])
def test_gloo_sharded_steps_in_two_halves(tmp_path, shard_rows, bad_rank):
    """`step_begin` / `finish` of both matchers (what bench.py's N > 1 step runs on): two steps enqueued back to back, finished
    afterwards, equal to the oracle on the unsharded bank.  With `bad_rank` one shard's lists are PROVISIONAL for 7 queries
    (wrong rows, winning scores) until its search is finished: its count travels with the lists, every rank sees it and all
    of them redo exchange and merge with the re-scanned lists -- nothing provisional survives."""
    from helpers import unit_rows
    from oracle import pyoracle
    world, port = len(shard_rows), 31500 + (os.getpid() % 1000) + 13 * len(shard_rows) + bad_rank
    mp.spawn(_row_worker_async, args=(world, port, str(tmp_path), shard_rows, bad_rank), nprocs=world, join=True)
    offs = np.concatenate(([0], np.cumsum(shard_rows)))
    whole = unit_rows(np.random.default_rng(99), sum(shard_rows), 64)
    for r in range(world):
        got = np.load(tmp_path / f"a{r}.npz")
        for t in range(2):
            q = unit_rows(np.random.default_rng(4321 + 10 * t + r), 40, 64)
            i, s, c = pyoracle.nns_search(whole, q, 5)
            assert np.array_equal(got[f"rows{t}"], i) and np.array_equal(got[f"sims{t}"], s)
        assert np.array_equal(got["cnt0"], c * 0 + 5)
        shard = whole[offs[r]:offs[r + 1]]
        q = unit_rows(np.random.default_rng(4321 + r), 40, 64)
        assert np.array_equal(got["intra_rows"], pyoracle.nns_search(shard, q, 5)[0])
        oq = np.concatenate([unit_rows(np.random.default_rng(4321 + o), 40, 64) for o in range(world) if o != r])
        i, s, _ = pyoracle.nns_search(shard, oq, 1)
        assert np.array_equal(got["inter_rows"], i) and np.array_equal(got["inter_sims"], s)


def test_host_merge_stand_in_order():
    rows = np.array([[[2, 0, -1]], [[1, 0, -1]]])                      # shard 0: rows 2,0 ; shard 1: rows 1,0
    sims = np.array([[[0.5, 0.25, np.nan]], [[np.nan, 0.5, np.nan]]])   # shard 1 starts with a NaN score
    r, s, c = _host_merge(rows, sims, np.array([[2], [2]]), [0, 10])
    assert r.tolist() == [[11, 10, 2]] and c.tolist() == [3]            # NaN first, tie 0.5 -> larger global row
    assert np.isnan(s[0, 0]) and s[0, 1:].tolist() == [0.5, 0.5]


def test_empty_step_returns_empty_results_without_a_collective():
    """m == 0 (equal on all ranks): both matchers return empty results; no gather / search / exchange is issued
    (the one-gather code did, the chunked code raised on view(world, 0, -1))."""
    from cslam_amd.sharded import RowShardedBankMatcher, ShardedInterRobotMatcher

    def never(*a, **kw):
        raise AssertionError("an empty step must not reach the search / the collectives")
    m = ShardedInterRobotMatcher(1, 2, never, k_intra=5, gather_fn=never, chunks=2)
    intra, inter = m.step(torch.zeros(0, 16))
    assert intra[0].shape == (0, 5) and intra[1].shape == (0, 5) and intra[2].shape == (0,)
    assert inter[0].shape == (0, 1) and inter[1].shape == (0, 1) and inter[2].shape == (0,) and inter[3].shape == (0,)
    assert intra[0].dtype == torch.int64 and intra[1].dtype == torch.float64 and intra[2].dtype == torch.int32
    r = RowShardedBankMatcher(0, 2, never, [0, 50], k=5, gather_fn=never, exchange_fn=never, merge_fn=never, chunks=3)
    rows, sims, cnt = r.step(torch.zeros(0, 16))
    assert rows.shape == (0, 5) and sims.shape == (0, 5) and cnt.shape == (0,)
    assert rows.dtype == torch.int64 and sims.dtype == torch.float64 and cnt.dtype == torch.int32
