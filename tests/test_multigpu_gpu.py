"""GPU suite, armed only where >= 2 GPUs are visible (skipped on the 1-GPU build box, run wherever a multi-GPU node
executes `pytest -m gpu`): the multi-GPU path over REAL RCCL, one process per GPU.

  * the C ABI's exchange (`cslam_comm_init`, `cslam_allgather_queries_dev`, `cslam_exchange_lists_dev`, csrc/comm.hip) with
    two ranks: the id travels through a file, every byte of both collectives is checked;
  * `RowShardedBankMatcher` (ONE bank split by rows) and `ShardedInterRobotMatcher` (one robot bank per GPU,
    the reference's per-bank best-1 semantics, cslam/loop_closure_sparse_matching.py:45-53,56-72) over torch.distributed's
    "nccl" backend (= RCCL): HIP search of the shard, RCCL all-gather / all-to-all, HIP merge -- against the CPU oracle on
    the UNSHARDED bank, indices identical, float64 scores within 1e-12.

The gloo twins of the same control flow run on CPU in tests/test_sharded_cpu.py; two ranks sharing one GPU (collectives
staged through gloo) in tests/test_sharded_gpu.py."""
import os
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

N_GPUS = torch.cuda.device_count() if torch.cuda.is_available() else 0
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(N_GPUS < 2, reason="needs >= 2 GPUs (RCCL between devices); %d visible" % N_GPUS)]

N_ROWS, DIM, M, K = 9000, 512, 300, 5


def _cuts(world):
    return [g * N_ROWS // world + (37 if 0 < g < world else 0) for g in range(world + 1)]    # uneven shards


def _env(rank, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)


def _rccl_worker(rank, world, port, outdir):
    _env(rank, port)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from cslam_amd import nns_matching as nnm
    from cslam_amd.sharded import RowShardedBankMatcher, ShardedInterRobotMatcher
    from helpers import unit_rows
    cuts = _cuts(world)
    whole = unit_rows(np.random.default_rng(77), N_ROWS, DIM)
    dev = torch.device("cuda", rank)
    # (i) ONE bank split by rows
    nn = nnm.NearestNeighborsMatching(device=rank)
    nn.add_items(whole[cuts[rank]:cuts[rank + 1]], range(cuts[rank + 1] - cuts[rank]))
    local = torch.from_numpy(unit_rows(np.random.default_rng(500 + rank), M, DIM)).to(dev)
    out = {}
    for chunks in (1, 3):
        m = RowShardedBankMatcher(rank, world, lambda q, k: nn.search_device(q, k, mode=nnm.MODE_MFMA), cuts, k=K, chunks=chunks)
        rows, sims, cnt = m.step(local)
        torch.cuda.synchronize()
        out["rows%d" % chunks], out["sims%d" % chunks], out["cnt%d" % chunks] = rows.cpu().numpy(), sims.cpu().numpy(), cnt.cpu().numpy()
    assert nn.last_stats()[1] == nnm.MODE_MFMA
    # (ii) one robot bank per GPU
    own = unit_rows(np.random.default_rng(1234 + rank), 2000, DIM)
    nr = nnm.NearestNeighborsMatching(device=rank)
    nr.add_items(own, range(2000))
    mr = ShardedInterRobotMatcher(rank, world, lambda q, k: nr.search_device(q, k, mode=nnm.MODE_MFMA), k_intra=K, chunks=2)
    intra, inter = mr.step(local)
    torch.cuda.synchronize()
    out.update(intra_rows=intra[0].cpu().numpy(), intra_sims=intra[1].cpu().numpy(), inter_rows=inter[0].cpu().numpy(),
               inter_sims=inter[1].cpu().numpy(), inter_robot=inter[3].cpu().numpy())
    # (iii) the two-half step over RCCL with a CLUSTERED shard: near-duplicates inside the fp16 candidate stage's re-scoring window
    # leave queries uncertified, so finish() takes its second exchange + merge with the re-scanned lists -- and it is called on a
    # DIFFERENT stream than step_begin() (the lists must be ordered behind the re-scan whatever stream the caller finishes on)
    base = unit_rows(np.random.default_rng(4242), 40, DIM)
    rng = np.random.default_rng(31 + rank)
    clustered = np.repeat(base, (cuts[rank + 1] - cuts[rank] + 39) // 40, axis=0)[:cuts[rank + 1] - cuts[rank]]
    clustered = clustered + 2e-4 * rng.standard_normal(clustered.shape).astype(np.float32)
    clustered /= np.linalg.norm(clustered, axis=1, keepdims=True)
    ncl = nnm.NearestNeighborsMatching(device=rank)
    ncl.add_items(clustered.astype(np.float32), range(clustered.shape[0]))
    qcl = torch.from_numpy(np.tile(base, (M // 40 + 1, 1))[:M].astype(np.float32)).to(dev)
    mc = RowShardedBankMatcher(rank, world, lambda q, k: ncl.search_device(q, k, mode=nnm.MODE_MFMA), cuts, k=K, chunks=1,
                               search_async_fn=lambda q, k: ncl.search_device_async(q, k, mode=nnm.MODE_MFMA))
    pend = mc.step_begin(qcl)
    other = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(other):
        rows, sims, cnt = pend.finish()
        rows_h, sims_h, cnt_h = rows.cpu(), sims.cpu(), cnt.cpu()            # reads on the caller's stream
    torch.cuda.synchronize()
    out.update(cl_rows=rows_h.numpy(), cl_sims=sims_h.numpy(), cl_cnt=cnt_h.numpy(), cl_uncertified=np.array([ncl.last_stats()[0]]))
    np.savez(os.path.join(outdir, "cl_bank%d.npz" % rank), bank=clustered.astype(np.float32))
    np.savez(os.path.join(outdir, "g%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", sorted({2, min(N_GPUS, 8)} if N_GPUS >= 2 else {2}))
def test_rccl_sharded_matchers_equal_the_oracle_on_the_unsharded_bank(tmp_path, world):
    from helpers import assert_topk_equal, unit_rows
    from oracle import pyoracle
    port = 32500 + (os.getpid() % 1000) + world
    mp.spawn(_rccl_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    whole = unit_rows(np.random.default_rng(77), N_ROWS, DIM)
    qs = [unit_rows(np.random.default_rng(500 + r), M, DIM) for r in range(world)]
    for r in range(world):
        got = np.load(tmp_path / ("g%d.npz" % r))
        oi, os_, oc = pyoracle.nns_search(whole, qs[r], K)
        for chunks in (1, 3):
            assert_topk_equal(got["rows%d" % chunks], got["sims%d" % chunks], got["cnt%d" % chunks], oi, os_, oc, 1e-12)
        own = unit_rows(np.random.default_rng(1234 + r), 2000, DIM)
        i, s, _ = pyoracle.nns_search(own, qs[r], K)
        assert np.array_equal(got["intra_rows"], i) and np.max(np.abs(got["intra_sims"] - s)) <= 1e-12
        others = [o for o in range(world) if o != r]
        i, s, _ = pyoracle.nns_search(own, np.concatenate([qs[o] for o in others]), 1)
        assert np.array_equal(got["inter_rows"], i) and np.max(np.abs(got["inter_sims"] - s)) <= 1e-12
        assert np.array_equal(got["inter_robot"], np.repeat(others, M))
    # the clustered shards: every rank's result equals the oracle on the concatenated clustered bank, and the redo path ran
    whole_cl = np.concatenate([np.load(tmp_path / ("cl_bank%d.npz" % r))["bank"] for r in range(world)])
    base = unit_rows(np.random.default_rng(4242), 40, DIM)
    qcl = np.tile(base, (M // 40 + 1, 1))[:M].astype(np.float32)
    oi, os_, oc = pyoracle.nns_search(whole_cl, qcl, K)
    redo = 0
    for r in range(world):
        got = np.load(tmp_path / ("g%d.npz" % r))
        assert_topk_equal(got["cl_rows"], got["cl_sims"], got["cl_cnt"], oi, os_, oc, 1e-12)
        redo += int(got["cl_uncertified"][0])
    assert redo > 0, "the clustered bank left no query uncertified: the second exchange was not exercised"


def _cabi_worker(rank, world, outdir):
    import ctypes as C
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    from cslam_amd import _lib
    lib = _lib.load()
    idfile = os.path.join(outdir, "rccl_id.bin")
    ident = (C.c_char * 128)()
    if rank == 0:                                          # "whatever channel the host has": a file
        _lib.check(lib.cslam_comm_unique_id(ident))
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(ident))
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            assert time.time() - t0 < 120, "rank 0 never published the communicator id"
            time.sleep(0.05)
        ident = (C.c_char * 128).from_buffer_copy(open(idfile, "rb").read())
    comm = C.c_void_p()
    _lib.check(lib.cslam_comm_init(world, rank, ident, rank, C.byref(comm)))
    w, r = C.c_int(-1), C.c_int(-1)
    _lib.check(lib.cslam_comm_info(comm, C.byref(w), C.byref(r)))
    assert (w.value, r.value) == (world, rank)
    dev = torch.device("cuda", rank)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rows, d = 33, 4096
    q = torch.from_numpy(np.random.default_rng(900 + rank).standard_normal((rows, d)).astype(np.float32)).to(dev)
    allq = torch.zeros((world * rows, d), device=dev)
    _lib.check(lib.cslam_allgather_queries_dev(comm, C.c_void_p(q.data_ptr()), rows, d * 4, C.c_void_p(allq.data_ptr()), st))
    # lists[dst] = what this rank sends to rank dst: value encodes (src, dst, position)
    w11 = 2 * K + 1
    lists = torch.from_numpy((np.arange(world * rows * w11, dtype=np.int64).reshape(world, rows, w11) + 10 ** 9 * (rank + 1))).to(dev)
    got = torch.zeros_like(lists)
    _lib.check(lib.cslam_exchange_lists_dev(comm, C.c_void_p(lists.data_ptr()), C.c_void_p(got.data_ptr()), rows * w11 * 8, st))
    torch.cuda.synchronize()
    want_q = np.concatenate([np.random.default_rng(900 + g).standard_normal((rows, d)).astype(np.float32) for g in range(world)])
    assert np.array_equal(allq.cpu().numpy(), want_q), "all-gather: rank-major concatenation of every rank's rows"
    base = np.arange(world * rows * w11, dtype=np.int64).reshape(world, rows, w11)
    want_l = np.stack([base[rank] + 10 ** 9 * (s + 1) for s in range(world)])        # slice s = what rank s sent to me
    assert np.array_equal(got.cpu().numpy(), want_l), "all-to-all: slice s of the result came from rank s"
    _lib.check(lib.cslam_comm_destroy(comm))
    open(os.path.join(outdir, "ok%d" % rank), "w").write("ok")


def test_c_abi_exchange_two_ranks_over_rccl(tmp_path):
    """csrc/comm.hip with a real 2-rank communicator (tests/test_sharded_gpu.py can only form a 1-rank one)."""
    world = 2
    mp.spawn(_cabi_worker, args=(world, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


@pytest.mark.parametrize("mode", ["rows", "robots"])
def test_bench_self_launch_over_rccl(mode):
    """`python bench.py --gpus N` on a multi-GPU box: launches its own N ranks over RCCL, prints ONE JSON line with n_gpus = N, and in
    rows mode the sharded step equals a single-GPU search of the whole (one, seeded) bank on every rank (`sharded_check`)."""
    import json
    import subprocess
    n = min(N_GPUS, 8)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--bank-rows", "40000", "--batch", "512", "--match-queries", "4096", "--shard-mode", mode],
                       capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["ranks"] == n and d["collective_backend"] == "nccl (RCCL)" and d["value"] > 0
    if mode == "rows":
        assert d["sharded_check"]["equal_to_unsharded_bank_on_every_rank"] is True
