"""CPU suite: the N > 1 control flow (one robot bank per rank, all-gather of new descriptors,
local top-k / best-1) on 2 gloo ranks, with the CPU oracle injected as the search function."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, outdir):
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

    m = ShardedInterRobotMatcher(rank, world, search, k_intra=5)
    local = torch.from_numpy(unit_rows(np.random.default_rng(4321 + rank), 40, 64))
    intra, inter = m.step(local)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), intra_rows=intra[0].numpy(), intra_sims=intra[1].numpy(),
             inter_rows=inter[0].numpy(), inter_sims=inter[1].numpy(), inter_robot=inter[3].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_step(tmp_path):
    from helpers import unit_rows
    from oracle import pyoracle
    world, port = 2, 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    banks = [unit_rows(np.random.default_rng(1234 + r), 500, 64) for r in range(world)]
    qs = [unit_rows(np.random.default_rng(4321 + r), 40, 64) for r in range(world)]
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npz")
        i, s, _ = pyoracle.nns_search(banks[r], qs[r], 5)           # own keyframes vs own bank
        assert np.array_equal(got["intra_rows"], i) and np.array_equal(got["intra_sims"], s)
        o = 1 - r
        i, s, _ = pyoracle.nns_search(banks[r], qs[o], 1)           # the other robot's keyframes vs own bank
        assert np.array_equal(got["inter_rows"], i) and np.array_equal(got["inter_sims"], s)
        assert np.all(got["inter_robot"] == o)


def test_single_rank_is_passthrough():
    from cslam_amd.sharded import ShardedInterRobotMatcher
    calls = []

    def search(q, k):
        calls.append((tuple(q.shape), k))
        return q, q, q

    m = ShardedInterRobotMatcher(0, 1, search, k_intra=7)
    intra, inter = m.step(torch.zeros(3, 8))
    assert inter is None and calls == [((3, 8), 7)]
