"""GPU suite: the multi-process row-sharded bank end to end on ONE GPU box -- 2 ranks share cuda:0, every rank runs the
product path (HIP search of its shard through the C ABI, HIP merge kernel); only the two collectives are staged
through gloo/CPU because a 1-GPU box has no second device for RCCL.  Result: the oracle's top-k over the whole bank."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu

N_ROWS, DIM, M, K = 6000, 512, 300, 5
CUTS = (0, 3500, 6000)


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cslam_amd import nns_matching as nnm
    from cslam_amd.sharded import RowShardedBankMatcher
    from helpers import unit_rows
    whole = unit_rows(np.random.default_rng(77), N_ROWS, DIM)
    nn = nnm.NearestNeighborsMatching()
    nn.add_items(whole[CUTS[rank]:CUTS[rank + 1]], range(CUTS[rank + 1] - CUTS[rank]))

    def gather(local, w, group=None):
        out = torch.empty((w * local.shape[0], local.shape[1]), dtype=local.dtype)
        dist.all_gather_into_tensor(out, local.cpu().contiguous())
        return out.cuda()

    def exchange(packed, w, group=None):
        out = torch.empty(packed.shape, dtype=packed.dtype)
        dist.all_to_all_single(out, packed.cpu().contiguous())
        return out.cuda()

    m = RowShardedBankMatcher(rank, world, lambda q, k: nn.search_device(q, k, mode=nnm.MODE_MFMA), CUTS, k=K,
                              gather_fn=gather, exchange_fn=exchange)
    local = torch.from_numpy(unit_rows(np.random.default_rng(500 + rank), M, DIM)).cuda()
    rows, sims, cnt = m.step(local)
    torch.cuda.synchronize()
    assert nn.last_stats()[1] == nnm.MODE_MFMA
    np.savez(os.path.join(outdir, f"g{rank}.npz"), rows=rows.cpu().numpy(), sims=sims.cpu().numpy(), cnt=cnt.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_row_sharded_bank_on_one_gpu(tmp_path):
    from helpers import assert_topk_equal, unit_rows
    from oracle import pyoracle
    world, port = 2, 31500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    whole = unit_rows(np.random.default_rng(77), N_ROWS, DIM)
    for r in range(world):
        q = unit_rows(np.random.default_rng(500 + r), M, DIM)
        oi, os_, oc = pyoracle.nns_search(whole, q, K)
        got = np.load(tmp_path / f"g{r}.npz")
        assert_topk_equal(got["rows"], got["sims"], got["cnt"], oi, os_, oc, 1e-12)


def test_c_abi_exchange_single_rank():
    """cslam_comm_* / cslam_allgather_queries_dev / cslam_exchange_lists_dev (csrc/comm.hip): the C-ABI twin of the two
    collectives of cslam_amd/sharded.py.  A 1-GPU box can only form a one-rank RCCL communicator, where both collectives
    are copies; the N-rank forms run on the driver's 8-GPU node through bench.py (torch.distributed, same RCCL)."""
    import ctypes as C
    import torch
    from cslam_amd import _lib
    lib = _lib.load()
    ident = (C.c_char * 128)()
    _lib.check(lib.cslam_comm_unique_id(ident))
    comm = C.c_void_p()
    _lib.check(lib.cslam_comm_init(1, 0, ident, 0, C.byref(comm)))
    w, r = C.c_int(-1), C.c_int(-1)
    _lib.check(lib.cslam_comm_info(comm, C.byref(w), C.byref(r)))
    assert (w.value, r.value) == (1, 0)
    q = torch.randn((33, 4096), device="cuda")
    allq = torch.zeros_like(q)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.cslam_allgather_queries_dev(comm, C.c_void_p(q.data_ptr()), 33, 4096 * 4, C.c_void_p(allq.data_ptr()), st))
    lists = torch.randint(0, 2 ** 40, (1, 33, 11), device="cuda", dtype=torch.int64)
    got = torch.zeros_like(lists)
    _lib.check(lib.cslam_exchange_lists_dev(comm, C.c_void_p(lists.data_ptr()), C.c_void_p(got.data_ptr()), 33 * 11 * 8, st))
    torch.cuda.synchronize()
    assert torch.equal(allq, q) and torch.equal(got, lists)
    _lib.check(lib.cslam_comm_destroy(comm))
    # wrong rank is refused before RCCL is asked
    assert lib.cslam_comm_init(2, 2, ident, 0, C.byref(comm)) != 0
