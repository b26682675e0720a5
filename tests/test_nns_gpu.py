"""GPU parity tests of the descriptor-matching hot path (run with -m gpu on an MI355X).

Every test goes Python class -> ctypes -> libcslam_hip.so (C ABI) -> HIP kernels and compares
with (a) the reference-generated golden vectors and (b) the CPU oracle on identical inputs.
Bar: top-k indices bit-identical; scores within 1e-5 of the reference (north_star gate) and
within 1e-12 of the float64 oracle.
"""
import os

import numpy as np
import pytest

from helpers import GOLDEN, assert_topk_equal, nns_case_inputs, nns_case_names, unit_rows
from oracle import pyoracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nnm():
    from cslam_amd import nns_matching
    return nns_matching


def make_bank(nnm, bank):
    nn = nnm.NearestNeighborsMatching()
    nn.add_items(bank, range(bank.shape[0]))
    return nn


@pytest.mark.parametrize("mode", ["scan", "mfma"])
def test_golden_reference_vectors(nnm, mode):
    g = np.load(GOLDEN + "/nns_g1.npz")
    m = nnm.MODE_SCAN if mode == "scan" else nnm.MODE_MFMA
    for name in nns_case_names(g):
        bank, q = nns_case_inputs(g, name)
        k = int(g[name + "/k"])
        nn = make_bank(nnm, bank)
        idx, sims, cnt = nn.search_batch(q, k, mode=m)
        assert_topk_equal(idx, sims, cnt, g[name + "/idx"], g[name + "/sims"], g[name + "/cnt"], 1e-5)
        oi, os_, oc = pyoracle.nns_search(bank, q, k)
        assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
        if mode == "mfma" and k <= 8:
            assert nn.last_stats()[1] == nnm.MODE_MFMA


def test_reference_single_query_api(nnm):
    """add_item / search / search_best exactly as the reference's callers use them."""
    g = np.load(GOLDEN + "/nns_g1.npz")
    name = "r_n257_d512_k10_f32"
    bank, q = nns_case_inputs(g, name)
    nn = nnm.NearestNeighborsMatching()
    assert nn.search(q[0], 3) == ([], []) and nn.search_best(q[0]) == (None, None)
    for i in range(bank.shape[0]):
        nn.add_item(bank[i], 1000 + i)          # items are arbitrary ids
    assert nn.n == 257 and nn.dim == 512
    assert np.array_equal(nn.data[:nn.n], bank) and nn.data.shape == (1000, 512)
    for j in range(4):
        items, sims = nn.search(q[j], 10)
        assert items == [1000 + int(r) for r in g[name + "/idx"][j]]
        assert np.max(np.abs(sims - g[name + "/sims"][j])) < 1e-5
        best, s = nn.search_best(q[j])
        assert best == items[0] and s == sims[0]


@pytest.mark.parametrize("n,d,nq,k", [(5000, 4096, 300, 5), (1537, 512, 130, 8), (777, 100, 257, 3),
                                      (300, 10, 129, 1), (129, 64, 1000, 5)])
def test_mfma_vs_oracle_seeded(nnm, n, d, nq, k):
    bank = unit_rows(np.random.default_rng(n + d), n, d)
    q = unit_rows(np.random.default_rng(n + d + 1), nq, d)
    nn = make_bank(nnm, bank)
    oi, os_, oc = pyoracle.nns_search(bank, q, k)
    for mode in (nnm.MODE_MFMA, nnm.MODE_SCAN, nnm.MODE_AUTO):
        idx, sims, cnt = nn.search_batch(q, k, mode=mode)
        assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
    # float64 queries (the inter-robot path: lcsm.py:63 np.asarray(msg.descriptor))
    qd = np.random.default_rng(9).standard_normal((64, d))
    oi, os_, oc = pyoracle.nns_search(bank, qd, k)
    for mode in (nnm.MODE_MFMA, nnm.MODE_SCAN):
        idx, sims, cnt = nn.search_batch(qd, k, mode=mode)
        assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)


def test_non_unit_norm_and_scaling(nnm):
    rng = np.random.default_rng(3)
    bank = (rng.standard_normal((900, 96)) * rng.uniform(1e-3, 1e3, size=(900, 1))).astype(np.float32)
    q = (rng.standard_normal((200, 96)) * 7.0).astype(np.float32)
    nn = make_bank(nnm, bank)
    oi, os_, oc = pyoracle.nns_search(bank, q, 5)
    for mode in (nnm.MODE_MFMA, nnm.MODE_SCAN):
        idx, sims, cnt = nn.search_batch(q, 5, mode=mode)
        assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)


@pytest.mark.parametrize("mode", ["scan", "mfma"])
def test_causal_row_limit(nnm, mode):
    """Row j sees rows < j: the intra-robot order of gdlcd.py:157-160 done as one batch."""
    m = nnm.MODE_SCAN if mode == "scan" else nnm.MODE_MFMA
    bank = unit_rows(np.random.default_rng(21), 700, 256)
    nn = make_bank(nnm, bank)
    lim = np.arange(700, dtype=np.int64)
    idx, sims, cnt = nn.search_batch(bank, 5, row_limit=lim, mode=m)
    oi, os_, oc = pyoracle.nns_search(bank, bank, 5, row_limit=lim)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
    assert cnt[0] == 0 and cnt[3] == 3 and np.all(idx[0] == -1)


def test_duplicates_zero_rows_and_ties(nnm):
    rng = np.random.default_rng(4)
    bank = rng.standard_normal((400, 128)).astype(np.float32)
    bank[100] = bank[7]; bank[250] = bank[7]; bank[399] = bank[7]      # exact ties -> larger row first
    bank[33] = 0.0; bank[301] = 0.0                                     # NaN scores rank first
    q = np.concatenate([bank[7:8], rng.standard_normal((140, 128)).astype(np.float32)])
    nn = make_bank(nnm, bank)
    oi, os_, oc = pyoracle.nns_search(bank, q, 8)
    assert list(oi[0, :6]) == [301, 33, 399, 250, 100, 7]
    for mode in (nnm.MODE_MFMA, nnm.MODE_SCAN):
        idx, sims, cnt = nn.search_batch(q, 8, mode=mode)
        assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)


def test_certificate_fallback_on_unresolvable_near_ties(nnm):
    """Rows that differ by less than float32 roundoff of the dot product: the fp32-MFMA stage
    cannot rank them, the certificate must fail, and the float64 scan must still give the
    oracle's exact order."""
    rng = np.random.default_rng(6)
    base = unit_rows(rng, 1, 1024)[0]
    bank = np.tile(base, (600, 1)).astype(np.float32)
    # perturb one coordinate per row by a few float32 ulps -> cosine differences ~1e-9
    for i in range(600):
        bank[i, i % 1024] = np.nextafter(bank[i, i % 1024], np.float32(1.0)) if i % 3 else bank[i, i % 1024]
        bank[i, (7 * i) % 1024] *= np.float32(1.0 + (i % 5) * 1.2e-7)
    q = unit_rows(rng, 130, 1024)
    nn = make_bank(nnm, bank)
    idx, sims, cnt = nn.search_batch(q, 5, mode=nnm.MODE_MFMA)
    assert nn.last_stats()[0] > 0, "certificate should have rejected these queries"
    oi, os_, oc = pyoracle.nns_search(bank, q, 5)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)


def test_large_k_multi_pass(nnm):
    """k >= n and k > 64 (the reference test uses k = n = 100: test_sparse_matching.py:62)."""
    bank = unit_rows(np.random.default_rng(8), 100, 100)
    q = unit_rows(np.random.default_rng(9), 5, 100)
    nn = make_bank(nnm, bank)
    for k in (100, 103, 65, 64):
        idx, sims, cnt = nn.search_batch(q, k)
        oi, os_, oc = pyoracle.nns_search(bank, q, k)
        assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
    bank = unit_rows(np.random.default_rng(10), 3000, 64)
    nn = make_bank(nnm, bank)
    idx, sims, cnt = nn.search_batch(q[:, :64].copy(), 200)
    oi, os_, oc = pyoracle.nns_search(bank, q[:, :64].copy(), 200)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)


def test_growth_keeps_contents(nnm):
    """Amortised doubling (nns_matching.py:31-37): contents survive reallocation."""
    bank = unit_rows(np.random.default_rng(12), 2500, 48)
    nn = nnm.NearestNeighborsMatching()
    for s in range(0, 2500, 500):
        nn.add_items(bank[s:s + 500], range(s, s + 500))
    assert nn.n == 2500 and nn.data.shape == (4000, 48)
    assert np.array_equal(nn.data[:2500], bank)
    idx, sims, cnt = nn.search_batch(bank[:64], 1, mode=nnm.MODE_MFMA)
    assert np.array_equal(idx[:, 0], np.arange(64))


def test_full_size_bank_properties(nnm):
    """BASELINE config 3 size (100k x 4096 bank): size-independent properties.
    (1) a bank row queried against the bank finds itself first with similarity clipped to <= 1;
    (2) MFMA batch path == independent float64 scan path on the same queries;
    (3) a few queries against the CPU oracle (bit-identical indices)."""
    import torch
    n, d = 100_000, 4096
    gen = torch.Generator(device="cuda").manual_seed(1234)
    bank = torch.randn((n, d), generator=gen, device="cuda", dtype=torch.float32)
    bank /= bank.norm(dim=1, keepdim=True)
    nn = nnm.NearestNeighborsMatching()
    nn.add_items_device(bank)
    sel = torch.arange(0, n, 97, device="cuda")[:1024]
    q = bank[sel].contiguous()
    rows, sims, cnt = nn.search_device(q, 5, mode=nnm.MODE_MFMA)
    assert nn.last_stats()[1] == nnm.MODE_MFMA
    assert torch.equal(rows[:, 0], sel)
    assert float(sims[:, 0].min()) > 1 - 1e-6 and float(sims[:, 0].max()) <= 1.0
    assert torch.all(sims[:, :-1] >= sims[:, 1:])
    r2, s2, c2 = nn.search_device(q[:64].contiguous(), 5, mode=nnm.MODE_SCAN)
    assert torch.equal(rows[:64], r2) and torch.equal(cnt[:64], c2)
    assert float((sims[:64] - s2).abs().max()) < 1e-12
    hb = bank.cpu().numpy()
    oi, os_, oc = pyoracle.nns_search(hb, q[:6].cpu().numpy(), 5)
    assert np.array_equal(rows[:6].cpu().numpy(), oi)
    assert np.max(np.abs(sims[:6].cpu().numpy() - os_)) < 1e-12


@pytest.mark.parametrize("n,d,nq,k,f64", [(9000, 64, 8300, 5, False), (8200, 96, 8192, 8, True), (300_000, 64, 1024, 5, False)])
def test_large_tile_path_vs_oracle(nnm, n, d, nq, k, f64):
    """Large batches on the 256x256 MFMA tile (8 waves, per-lane lists of 8 with
    explicit drop bounds): ragged tile edges, causal limits, float64 queries.  300 000 rows against four query tiles: a patch
    column's share of the bank is longer than the 128 tiles the packed candidate lists of the persistent stage can address, so
    the schedule cuts the walk into two runs (csrc/sim_topk_ring.hip)."""
    bank = unit_rows(np.random.default_rng(n), n, d)
    q = unit_rows(np.random.default_rng(n + 1), nq, d)
    if f64:
        q = q.astype(np.float64) * 3.0
    nn = make_bank(nnm, bank)
    idx, sims, cnt = nn.search_batch(q, k, mode=nnm.MODE_MFMA)
    assert nn.last_stats()[1] == nnm.MODE_MFMA
    if not os.environ.get("CSLAM_MFMA_TILE"):
        assert nn.last_stats()[3] == -(-nq // 256)          # 256-row query tiles
    oi, os_, oc = pyoracle.nns_search(bank, q, k)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
    lim = np.minimum(np.arange(nq, dtype=np.int64) * 2, n)
    idx, sims, cnt = nn.search_batch(q, k, row_limit=lim, mode=nnm.MODE_MFMA)
    oi, os_, oc = pyoracle.nns_search(bank, q, k, row_limit=lim)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)


def test_large_tile_clustered_rows_force_bounds(nnm):
    """Near-duplicate clusters overflow the 8-entry per-lane lists of the 256 tile: the drop
    bound must flag those queries (or keep them certified correctly) -- results stay exact."""
    rng = np.random.default_rng(5)
    centers = unit_rows(rng, 40, 128)
    bank = np.repeat(centers, 210, axis=0) + 0.01 * rng.standard_normal((8400, 128)).astype(np.float32)
    bank = bank.astype(np.float32)
    q = np.repeat(centers, 208, axis=0)[:8200] + 0.01 * rng.standard_normal((8200, 128)).astype(np.float32)
    nn = make_bank(nnm, bank)
    idx, sims, cnt = nn.search_batch(q.astype(np.float32), 5, mode=nnm.MODE_MFMA)
    oi, os_, oc = pyoracle.nns_search(bank, q.astype(np.float32), 5)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)


@pytest.mark.parametrize("n,d,nq,k", [(3000, 256, 500, 10), (9000, 128, 8300, 16), (4000, 512, 300, 12)])
def test_mfma_path_serves_default_nb_best_matches(nnm, n, d, nq, k):
    """k up to 16 (the reference's default frontend.nb_best_matches is 10) stays on the MFMA path."""
    bank = unit_rows(np.random.default_rng(n + k), n, d)
    q = unit_rows(np.random.default_rng(n + k + 1), nq, d)
    nn = make_bank(nnm, bank)
    idx, sims, cnt = nn.search_batch(q, k, mode=nnm.MODE_AUTO)
    assert nn.last_stats()[1] == nnm.MODE_MFMA
    oi, os_, oc = pyoracle.nns_search(bank, q, k)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
    idx, sims, cnt = nn.search_batch(q[:40], 17, mode=nnm.MODE_AUTO)       # beyond the lists -> scan
    assert nn.last_stats()[1] == nnm.MODE_SCAN
    oi, os_, oc = pyoracle.nns_search(bank, q[:40], 17)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)


def test_empty_bank_with_known_dim(nnm):
    """A bank constructed with dim but no rows: every mode reports zero matches (the reference returns
    empty arrays from search, nns_matching.py:55-61 with n = 0)."""
    nn = nnm.NearestNeighborsMatching(dim=64)
    q = unit_rows(np.random.default_rng(0), 20, 64)
    for mode in (nnm.MODE_AUTO, nnm.MODE_SCAN, nnm.MODE_MFMA):
        idx, sims, cnt = nn.search_batch(q, 3, mode=mode)
        assert np.all(cnt == 0) and np.all(idx == -1) and np.all(np.isnan(sims))
    items, sims = nn.search(q[0], 5)
    assert items == [] and len(sims) == 0


@pytest.mark.parametrize("n,d,nq,k,f64", [(3000, 4096, 700, 5, False), (1111, 100, 333, 20, True),
                                         (130, 64, 70, 70, False), (5000, 512, 1500, 3, True)])
def test_exact_tile_kernel_many_queries_vs_oracle(nnm, n, d, nq, k, f64):
    """MODE_SCAN with >= 16 queries runs the 64-row x 64-query float64 tile kernel (also the certificate's fallback for
    many uncertified queries): ragged tiles, dims that are no multiple of the 64-column chunk, k > 64 lists, row limits."""
    rng = np.random.default_rng(n + nq)
    bank = unit_rows(rng, n, d)
    bank[n // 2] = bank[n // 3]                                   # an exact duplicate: tie broken by the larger row
    q = unit_rows(rng, nq, d)
    q[: nq // 4] = bank[rng.integers(0, n, nq // 4)] * np.float32(1.5)
    if f64:
        q = q.astype(np.float64)
    lim = rng.integers(0, n + 1, size=nq)
    nn = make_bank(nnm, bank)
    for row_limit in (None, lim):
        idx, sims, cnt = nn.search_batch(q, k, row_limit=row_limit, mode=nnm.MODE_SCAN)
        oi, os_, oc = pyoracle.nns_search(bank, q, k, row_limit=row_limit)
        assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)


def test_degenerate_descriptors_fall_back_without_a_cliff(nnm):
    """Near-duplicate descriptors (every cosine within 1e-6 of 1) defeat the fp32 certificate for most queries; the
    float64 tile kernel answers them exactly."""
    rng = np.random.default_rng(77)
    base = unit_rows(rng, 1, 1024)
    bank = (base + 2e-4 * rng.standard_normal((4000, 1024)).astype(np.float32)).astype(np.float32)
    q = (base + 2e-4 * rng.standard_normal((600, 1024)).astype(np.float32)).astype(np.float32)
    nn = make_bank(nnm, bank)
    idx, sims, cnt = nn.search_batch(q, 5, mode=nnm.MODE_MFMA)
    oi, os_, oc = pyoracle.nns_search(bank, q, 5)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
    assert nn.last_stats()[0] > 0                                 # the fallback really ran


@pytest.mark.parametrize("cuts,k", [((0, 150, 400), 8), ((0, 398, 400), 5), ((0, 100, 200, 300, 400), 1),
                                    ((0, 34, 35, 400), 16)])
def test_row_sharded_bank_merge_equals_whole_bank(nnm, cuts, k):
    """one bank split by rows over several GPUs.  Per-shard searches (here: several banks on one GPU)
    merged by cslam_topk_merge_dev give the whole-bank result bit for bit -- rows, float64 scores, counts --
    including exact ties across shards (larger global row first), NaN scores (first) and shards shorter than k.
    Every search runs the exact float64 kernel (MODE_SCAN) so that duplicate rows living in different shards get
    bit-identical scores: the scoring kernels agree to ~1e-16, not to the bit, and the order of EXACT duplicates
    across shards is only defined when one kernel scored them all (tie-free data: next test, MFMA mode)."""
    import torch
    from cslam_amd.sharded import merge_topk_device
    rng = np.random.default_rng(4)
    bank = rng.standard_normal((400, 128)).astype(np.float32)
    bank[100] = bank[7]; bank[250] = bank[7]; bank[399] = bank[7]      # exact ties in different shards
    bank[33] = 0.0; bank[301] = 0.0                                     # NaN scores in different shards
    q = np.concatenate([bank[7:8], rng.standard_normal((140, 128)).astype(np.float32)])
    oi, os_, oc = pyoracle.nns_search(bank, q, k)
    whole = make_bank(nnm, bank)
    dq = torch.from_numpy(q).cuda()
    wi, ws, wc = whole.search_device(dq, k, mode=nnm.MODE_SCAN)
    parts = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        parts.append(make_bank(nnm, bank[lo:hi]).search_device(dq, k, mode=nnm.MODE_SCAN))
    mi, ms, mc = merge_topk_device(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]),
                                   torch.stack([p[2] for p in parts]), cuts[:-1])
    torch.cuda.synchronize()
    assert_topk_equal(mi.cpu().numpy(), ms.cpu().numpy(), mc.cpu().numpy(), oi, os_, oc, 1e-12)
    assert torch.equal(mi, wi) and torch.equal(mc, wc)
    assert np.array_equal(ms.cpu().numpy().view(np.int64), ws.cpu().numpy().view(np.int64))   # scores bit for bit


def test_row_sharded_matcher_single_process(nnm):
    """RowShardedBankMatcher.step end to end on the GPU with the exchange replaced by an in-process transpose
    (2 simulated ranks on one device): each rank's keyframes get the oracle's top-k over the whole bank."""
    import torch
    from cslam_amd.sharded import RowShardedBankMatcher
    bank = unit_rows(np.random.default_rng(11), 3000, 256)
    offs = [0, 1700, 3000]
    shards = [make_bank(nnm, bank[offs[g]:offs[g + 1]]) for g in range(2)]
    qs = [torch.from_numpy(unit_rows(np.random.default_rng(20 + g), 64, 256)).cuda() for g in range(2)]
    allq = torch.cat(qs)
    packed = {}

    def run(g, sink):
        # chunks=1: the stand-in gather returns the whole step's descriptors whatever piece it is handed (the chunked,
        # overlapped form runs over real collectives in tests/test_sharded_cpu.py and tests/test_sharded_gpu.py)
        m = RowShardedBankMatcher(g, 2, lambda q, k: shards[g].search_device(q, k, mode=nnm.MODE_MFMA), offs, k=5,
                                  gather_fn=lambda local, w: allq, exchange_fn=sink, chunks=1)
        return m.step(qs[g])

    class _Stop(Exception):
        pass

    def capture(g):
        def sink(p, w):
            packed[g] = p.clone()
            raise _Stop()
        return sink
    for g in range(2):                                  # pass 1: record what every rank would send
        with pytest.raises(_Stop):
            run(g, capture(g))
    for g in range(2):                                  # pass 2: deliver slice g of every rank's buffer
        rows, sims, cnt = run(g, lambda p, w: torch.stack([packed[s][g] for s in range(2)]))
        oi, os_, oc = pyoracle.nns_search(bank, qs[g].cpu().numpy(), 5)
        assert_topk_equal(rows.cpu().numpy(), sims.cpu().numpy(), cnt.cpu().numpy(), oi, os_, oc, 1e-12)


def test_multi_bank_search_equals_separate_searches(nnm):
    """cslam_bank_search_multi_dev (a robot's local bank + its copies of the other robots' banks in one call, one host
    synchronisation): same rows / scores / counts as one cslam_bank_search_dev per bank and as the oracle -- banks of
    different sizes (one below the MFMA floor -> exact scan, one EMPTY but created), top-k with a causal row limit on
    the first bank, best-1 on the others, float32 and float64 queries, and a bank whose near-duplicate rows send
    queries through the certificate fallback AFTER the shared synchronisation."""
    import torch
    rng = np.random.default_rng(77)
    d = 256
    sizes = [3000, 120, 1500, 0, 700]
    banks_h = [unit_rows(np.random.default_rng(80 + i), max(n, 1), d)[:n] for i, n in enumerate(sizes)]
    base = unit_rows(rng, 1, d)[0]
    dup = np.tile(base, (700, 1)).astype(np.float32)                     # bank 4: unresolvable near-ties
    for i in range(700):
        dup[i, i % d] = np.nextafter(dup[i, i % d], np.float32(1.0)) if i % 3 else dup[i, i % d]
    banks_h[4] = dup
    banks = []
    for i, bh in enumerate(banks_h):
        nn = nnm.NearestNeighborsMatching(dim=d)
        if len(bh):
            nn.add_items(bh, range(len(bh)))
        banks.append(nn)
    m = 300
    for dtype in (np.float32, np.float64):
        q = unit_rows(rng, m, d).astype(dtype)
        qd = torch.from_numpy(q).cuda()
        lim = np.minimum(np.arange(m, dtype=np.int64) * 10, sizes[0])
        ks = [10, 1, 1, 1, 1]
        lims = [torch.from_numpy(lim).cuda(), None, None, None, None]
        got = nnm.search_multi_device(banks, qd, ks, lims)
        assert banks[4].last_stats()[0] > 0, "the near-duplicate bank should have needed the exact fallback"
        for i, nn in enumerate(banks):
            r, s, c = (t.cpu().numpy() for t in nn.search_device(qd, ks[i], row_limit=lims[i]))
            assert np.array_equal(got[i][0], r) and np.array_equal(got[i][2], c)
            assert np.array_equal(got[i][1], s, equal_nan=True)
            if sizes[i]:
                oi, os_, oc = pyoracle.nns_search(banks_h[i], q, ks[i], row_limit=lim if i == 0 else None)
                assert_topk_equal(got[i][0], got[i][1], got[i][2], oi, os_, oc, 1e-12)
            else:
                assert np.all(got[i][2] == 0) and np.all(got[i][0] == -1)


def test_multi_bank_search_is_ordered_after_the_producer_on_a_side_stream(nnm):
    """The per-bank streams of cslam_bank_search_multi_dev fork from the CALLER's stream: queries still being written by
    a long kernel chain on a non-blocking side stream (an extractor's output) must be complete before any bank reads them,
    and the caller's stream must own the results afterwards."""
    import torch
    d, m = 512, 600
    rng = np.random.default_rng(5)
    banks_h = [unit_rows(np.random.default_rng(200 + i), n, d) for i, n in enumerate((4000, 2500, 900, 3100))]
    banks = []
    for bh in banks_h:
        nn = nnm.NearestNeighborsMatching(dim=d)
        nn.add_items(bh, range(len(bh)))
        banks.append(nn)
    base = torch.from_numpy(unit_rows(rng, m, d)).cuda()
    mix = torch.from_numpy(np.linalg.qr(rng.standard_normal((d, d)))[0].astype(np.float32)).cuda()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(3):
        with torch.cuda.stream(side):
            q = base
            for _ in range(200):                                  # a few milliseconds of dependent kernels producing the queries
                q = q @ mix
            q = (q / q.norm(dim=1, keepdim=True)).contiguous()
            got = nnm.search_multi_device(banks, q, [5, 1, 1, 1])
        side.synchronize()
        qh = q.cpu().numpy()
        for i, bh in enumerate(banks_h):
            oi, os_, oc = pyoracle.nns_search(bh, qh, [5, 1, 1, 1][i])
            assert_topk_equal(got[i][0], got[i][1], got[i][2], oi, os_, oc, 1e-12)


def test_entry_points_leave_the_callers_device_alone(nnm):
    """Every C-ABI entry point runs on the device that owns its data and restores the caller's current device (a process
    may keep its banks on one GPU and run its extractor on another).  With one visible GPU this checks the guard is
    transparent; the cross-device half needs two."""
    import torch
    from cslam_amd.vpr import heads
    n_dev = torch.cuda.device_count()
    bank_dev = 1 if n_dev >= 2 else 0
    torch.cuda.set_device(0)
    bank = unit_rows(np.random.default_rng(5), 600, 128)
    nn = nnm.NearestNeighborsMatching(device=bank_dev)
    nn.add_items(bank, range(600))
    assert torch.cuda.current_device() == 0
    q = unit_rows(np.random.default_rng(6), 40, 128)
    idx, sims, cnt = nn.search_batch(q, 3)
    assert torch.cuda.current_device() == 0
    oi, os_, oc = pyoracle.nns_search(bank, q, 3)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
    qd = torch.from_numpy(q).to(f"cuda:{bank_dev}")
    r, s, c = nn.search_device(qd, 3)
    assert torch.cuda.current_device() == 0 and np.array_equal(r.cpu().numpy(), oi)
    # a head kernel on cuda:0 right after a bank call on the other device
    x = torch.randn((5, 64), device="cuda:0")
    y = heads.l2_normalize_(x.clone())
    assert y.device.index == 0 and torch.allclose(y.norm(dim=1), torch.ones(5, device="cuda:0"), atol=1e-5)
    assert nn.data.shape[1] == 128 and torch.cuda.current_device() == 0
    if n_dev < 2:
        pytest.skip("cross-device half needs 2 GPUs (the transparent half passed)")


def test_enqueue_finish_halves_equal_the_synchronous_search(nnm):
    """cslam_bank_search_enqueue_dev / _finish (one bank and a bank list): same results as the synchronous calls -- with a
    long-running kernel enqueued BETWEEN the two halves (the next step's extraction of a pipelining host: finish must not
    wait for it), with queries that fail the certificate (fallback enqueued by finish), and with the bank locked in between
    (add / a second search are refused, nothing is lost)."""
    import time
    import torch
    from cslam_amd._lib import CslamHipError
    rng = np.random.default_rng(91)
    d = 256
    bank_h = unit_rows(rng, 5000, d)
    nn = make_bank(nnm, bank_h)
    base = unit_rows(rng, 1, d)[0]
    dup = np.tile(base, (700, 1)).astype(np.float32)
    for i in range(700):
        dup[i, i % d] = np.nextafter(dup[i, i % d], np.float32(1.0)) if i % 3 else dup[i, i % d]
    nn_dup = make_bank(nnm, dup)
    q = unit_rows(rng, 400, d)
    qd = torch.from_numpy(q).cuda()
    want = [t.cpu().numpy() for t in nn.search_device(qd, 5, mode=nnm.MODE_MFMA)]
    filler = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    pend = nn.search_device_async(qd, 5, mode=nnm.MODE_MFMA)
    with pytest.raises(CslamHipError, match="not finished"):
        nn.add_items(bank_h[:3], range(3))
    with pytest.raises(CslamHipError, match="not finished"):
        nn.search_device(qd, 5)
    for _ in range(20):                                       # ~0.3 s of GPU work behind the search
        filler = filler @ filler
        filler /= filler.abs().max()
    t0 = time.perf_counter()
    got = pend.finish()
    waited = time.perf_counter() - t0
    busy = not torch.cuda.current_stream().query()            # the filler is still running: finish did not drain the stream
    assert pend.uncertified == 0 and pend.finish() is got     # idempotent
    got = [t.cpu().numpy() for t in got]
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    assert busy and waited < 0.15, (busy, waited)
    nn.add_items(bank_h[:3], range(5000, 5003))               # unlocked again
    assert nn.n == 5003
    # certificate failures: the fallback is enqueued by finish
    pend = nn_dup.search_device_async(qd, 5, mode=nnm.MODE_MFMA)
    r, s, c = (t.cpu().numpy() for t in pend.finish())
    assert pend.uncertified > 0
    oi, os_, oc = pyoracle.nns_search(dup, q, 5)
    assert_topk_equal(r, s, c, oi, os_, oc, 1e-12)
    # scan mode: nothing deferred, finish is a no-op
    pend = nn.search_device_async(qd[:4], 5)
    r, s, c = (t.cpu().numpy() for t in pend.finish())
    oi, os_, oc = pyoracle.nns_search(np.concatenate((bank_h, bank_h[:3])), q[:4], 5)
    assert_topk_equal(r, s, c, oi, os_, oc, 1e-12)
    # bank list, deferred: pinned result copies issued behind the search
    banks = [nn, nn_dup, make_bank(nnm, unit_rows(rng, 900, d))]
    ref = nnm.search_multi_device(banks, qd, [5, 1, 1])
    h = nnm.search_multi_device(banks, qd, [5, 1, 1], defer=True)
    for _ in range(5):
        filler = filler @ filler
        filler /= filler.abs().max()
    got = h.finish()
    for (a0, a1, a2), (b0, b1, b2) in zip(got, ref):
        assert np.array_equal(a0, b0) and np.array_equal(a1, b1, equal_nan=True) and np.array_equal(a2, b2)
    torch.cuda.synchronize()
    # the same bank twice in one list: refused (one workspace / side stream / pending slot per bank)
    with pytest.raises(CslamHipError, match="twice"):
        nnm.search_multi_device([nn, nn], qd, [5, 1])
    r, s, c = (t.cpu().numpy() for t in nn.search_device(qd, 5))      # and the bank is not left locked
    assert np.array_equal(r[:, 0], want[0][:, 0])


def _stage1_candidates(nnm, nn, nq):
    import ctypes as C
    nseg, eb = C.c_int(0), C.c_double(0.0)
    lib = nn._lib
    from cslam_amd import _lib
    _lib.check(lib.cslam_debug_last_candidates(nn._bank, nq, C.byref(nseg), None, None, C.byref(eb)))
    keys = np.empty((nq, nseg.value, 16), dtype=np.float32)
    rows = np.empty((nq, nseg.value, 16), dtype=np.int32)
    _lib.check(lib.cslam_debug_last_candidates(nn._bank, nq, C.byref(nseg), keys.ctypes.data_as(C.c_void_p),
                                               rows.ctypes.data_as(C.c_void_p), C.byref(eb)))
    return keys, rows, eb.value


@pytest.mark.parametrize("d,f64", [(4096, False), (4096, True), (512, False), (96, True)])
def test_fp16_pair_candidate_stage_error_is_inside_the_certificates_bound(nnm, monkeypatch, d, f64):
    """The candidate stages on the fp16 matrix pipe (csrc/sim_topk_pair.hip: "h1" = ONE product on the hi halves, the default;
    "pair" = exact hi / lo pairs, three products) are filters whose keys must stay within the bound handed to the float64
    certificate: |key - q.b/||b||| <= bound * ||q||.  Measured here on every candidate each stage kept (rows of very
    different magnitudes, values spread over twelve binades inside a row), against float64 -- and against the f32-input MFMA
    stage, whose bound is 5x tighter and whose results must be the same (all three are exact after stage 2)."""
    rng = np.random.default_rng(1000 + d)
    n, nq = 3000, 300
    bank = rng.standard_normal((n, d)).astype(np.float32)
    bank *= np.exp2(rng.integers(-12, 1, size=(n, d))).astype(np.float32)          # wide dynamic range inside a row
    bank *= np.exp2(rng.integers(-40, 40, size=(n, 1))).astype(np.float32)         # rows of very different norms
    q = rng.standard_normal((nq, d)) * np.exp2(rng.integers(-30, 30, size=(nq, 1)))
    q = q.astype(np.float64 if f64 else np.float32)
    nn = make_bank(nnm, bank)
    out = {}
    b64, q64 = bank.astype(np.float64), q.astype(np.float64)
    bn = np.sqrt((b64 * b64).sum(axis=1))
    qn = np.sqrt((q64 * q64).sum(axis=1))
    for stage in ("h1", "pair", "f32"):
        monkeypatch.setenv("CSLAM_MFMA_STAGE1", stage)
        idx, sims, cnt = nn.search_batch(q, 5, mode=nnm.MODE_MFMA)
        assert nn.last_stats()[1] == nnm.MODE_MFMA
        keys, rows, bound = _stage1_candidates(nnm, nn, nq)
        out[stage] = (idx, sims, cnt, bound)
        worst = 0.0
        for j in range(nq):
            r = rows[j].ravel()
            ok = r >= 0
            exact = (b64[r[ok]] @ q64[j]) / bn[r[ok]]
            worst = max(worst, float(np.max(np.abs(keys[j].ravel()[ok].astype(np.float64) - exact)) / qn[j]))
        assert worst <= bound, (stage, worst, bound)
        if stage == "pair":
            kd = (d + 31) // 32 * 32
            assert 3 * kd * 2.0 ** -23 < bound < 1.2 * (3 * kd + 64) * 2.0 ** -23 + 3e-6      # the pair bound: ~ 3 kd 2^-23
            assert worst < 2e-5, worst                          # and what the arithmetic really does: fp32-grade
        if stage == "h1":
            kh = (d + 63) // 64 * 64
            # the one-product bound: two fp16 roundings (2^-10, Cauchy-Schwarz) + kh fp32 additions
            # (+ one unit of the persistent stage's integer keys: 1.02 x 2^-16)
            assert 2.0 ** -10 + kh * 2.0 ** -23 < bound < 1.07 * (2.0 ** -10 + 1.01 * (kh + 64) * 2.0 ** -23) + 3e-6 + 1.02 * 2.0 ** -16
            assert worst < 2.0 ** -11, worst                    # what the arithmetic really does: random-sign fp16 roundings
    oi, os_, oc = pyoracle.nns_search(bank, q, 5)
    for stage in ("h1", "pair", "f32"):
        assert_topk_equal(out[stage][0], out[stage][1], out[stage][2], oi, os_, oc, 1e-12)
    assert out["pair"][3] > out["f32"][3] and out["h1"][3] > out["f32"][3]
    if d == 4096:
        assert abs(out["h1"][3] / out["pair"][3] - 1.0) < 0.01  # same window for stage 2: 1.582e-3 against 1.570e-3


def test_clustered_near_duplicates_stay_exact_and_back_off_to_the_f32_stage(nnm, monkeypatch):
    """Real place-recognition banks are clustered: revisits of a place are near-duplicates.  With hundreds of rows inside the fp16
    candidate stage's re-scoring window (2 x 1.57e-3 at 4096-D: more than the 64 contenders stage 2 re-scores) the certificate
    refuses and those queries go through the exact scan -- the results stay those of the oracle, bit for bit.  The bank notices
    (more than 1/32 of a search's queries uncertified) and runs its next searches on the f32-input stage, whose window is six
    times narrower; the fraction each stage leaves uncertified is what this test reports."""
    monkeypatch.delenv("CSLAM_MFMA_STAGE1", raising=False)
    rng = np.random.default_rng(77)
    d, n_c, per = 4096, 24, 150
    centres = unit_rows(rng, n_c, d)
    # rows of a cluster: centre + noise of norm a_r, a_r^2 uniform in [0, 5e-3]: cosine to the centre 1 - a_r^2 / 2, i.e. the 150 rows
    # spread evenly over 2.5e-3 -- all of them inside the fp16 stages' window (3.1e-3), about 30 inside the f32 stage's (5.2e-4)
    a = np.sqrt(rng.uniform(0.0, 5e-3, size=(n_c * per, 1))).astype(np.float32)
    bank = np.repeat(centres, per, axis=0) + a / np.sqrt(d) * rng.standard_normal((n_c * per, d)).astype(np.float32)
    bank = np.concatenate([bank.astype(np.float32), unit_rows(rng, 2000, d)])           # and unrelated rows around them
    q = (centres[rng.integers(0, n_c, size=400)] + (0.01 / np.sqrt(d)) * rng.standard_normal((400, d))).astype(np.float32)
    oi, os_, oc = pyoracle.nns_search(bank, q, 5)
    nn = make_bank(nnm, bank)
    frac = {}
    for tag in ("h1 (first search)", "after the back-off"):
        idx, sims, cnt = nn.search_batch(q, 5, mode=nnm.MODE_MFMA)
        assert nn.last_stats()[1] == nnm.MODE_MFMA
        assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
        _, _, bound = _stage1_candidates(nnm, nn, 4)
        frac[tag] = (nn.last_stats()[0] / 400.0, bound)
        # the stage that ran and the back-off are visible (cslam_bank_last_stage): fp16 stage, then 7 searches left of the first 8
        assert nn.last_stage() == ((1, 8, 16, False) if tag.startswith("h1") else (0, 7, 16, False))
    print("uncertified fraction / certificate bound per search:", frac)
    assert frac["h1 (first search)"][1] > 1e-3 and frac["h1 (first search)"][0] > 1.0 / 32      # the fp16 stage ran and gave up on many
    assert frac["after the back-off"][1] < 5e-4                                                 # the f32-input stage ran instead
    assert frac["after the back-off"][0] <= frac["h1 (first search)"][0]
    # the bank stays clustered: after the eight searches the fp16 stage is retried, overflows again, and the next back-off is twice as long
    for _ in range(7):
        nn.search_batch(q, 5, mode=nnm.MODE_MFMA)
    assert nn.last_stage() == (0, 0, 16, False)
    idx, sims, cnt = nn.search_batch(q, 5, mode=nnm.MODE_MFMA)
    assert nn.last_stage() == (1, 16, 32, False)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
    # an explicit choice of stage is never overridden, and says so
    monkeypatch.setenv("CSLAM_MFMA_STAGE1", "h1")
    idx, sims, cnt = nn.search_batch(q, 5, mode=nnm.MODE_MFMA)
    assert _stage1_candidates(nnm, nn, 4)[2] > 1e-3
    assert nn.last_stage()[0] == 1 and nn.last_stage()[3]
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)


@pytest.mark.parametrize("stage", ["h1", "pair", "f32"])
def test_search_beside_a_stream_that_thrashes_the_l2_is_bit_identical(nnm, monkeypatch, stage):
    """Every candidate stage stages its operands through LDS-DMA rings whose barriers wait on request counters: latency-dependent
    by construction.  The same search alone and beside a stream that streams 256 MB through the L2 over and over (operands that
    are L2 hits alone become HBM misses): rows, float64 scores and counts bit for bit."""
    import torch
    monkeypatch.setenv("CSLAM_MFMA_STAGE1", stage)
    rng = np.random.default_rng(31)
    bank = unit_rows(rng, 20000, 1024)
    q = unit_rows(rng, 3000, 1024)
    nn = make_bank(nnm, bank)
    ref = nn.search_batch(q, 5, mode=nnm.MODE_MFMA)
    big = torch.zeros(64 << 20, device="cuda")
    side = torch.cuda.Stream()
    for _ in range(4):
        with torch.cuda.stream(side):
            for _ in range(8):
                big.add_(1.0)
        got = nn.search_batch(q, 5, mode=nnm.MODE_MFMA)
        torch.cuda.synchronize()
        for a, b in zip(got, ref):
            assert np.array_equal(np.asarray(a), np.asarray(b))


def test_fp16_pair_stage_rows_and_queries_it_cannot_scale_are_still_exact(nnm):
    """Rows / queries whose magnitudes the power-of-two scale of the pair split cannot serve (beyond 2^+-100, non-finite
    entries) and zero rows: the candidate stage marks them (NaN key -> always a contender / query uncertified) and the
    float64 stages give them the score the reference gives them."""
    rng = np.random.default_rng(2024)
    n, d, nq = 2500, 256, 280
    bank = unit_rows(rng, n, d)
    bank[7] *= np.float32(1e-35)          # tiny row: cosine is scale-free, the reference ranks it like any other
    bank[8] *= np.float32(3e35)           # huge row
    bank[9] = 0.0                         # zero row: NaN similarity, ranks first (reference: argsort()[::-1])
    bank[10, 3] = np.inf                  # non-finite row: NaN similarity
    q = unit_rows(rng, nq, d)
    q[0] = bank[7] * np.float32(1e30)     # best match of q[0] is the tiny row
    q[1] = bank[8] * np.float32(1e-36)
    q[2] *= np.float32(1e-38)             # query below the servable range
    q[3] *= np.float32(1e37)
    nn = make_bank(nnm, bank)
    idx, sims, cnt = nn.search_batch(q, 5, mode=nnm.MODE_MFMA)
    assert nn.last_stats()[1] == nnm.MODE_MFMA
    oi, os_, oc = pyoracle.nns_search(bank, q, 5)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
    assert 7 in idx[0] and 8 in idx[1]


def test_packed_candidate_lists_rank_negative_and_tied_keys_like_the_oracle(nnm):
    """The persistent candidate stage keeps its per-lane lists as 32-bit integers, -key << 13 | position (csrc/sim_topk_pair_dev.h):
    queries whose best similarities are all NEGATIVE (the integer key changes sign), banks of exact duplicates (hundreds of rows with
    the same integer key: only the position bits order them), a zero query (unit of the keys from a zero norm) and row limits that
    leave fewer rows than k -- all through the 256 x 256 tiles (>= 257 queries), all equal to the oracle."""
    rng = np.random.default_rng(77)
    n, d, nq, k = 6000, 128, 300, 6
    u = unit_rows(rng, 1, d)[0]
    bank = (-u[None, :] * rng.uniform(0.5, 2.0, size=(n, 1)) + 0.02 * rng.standard_normal((n, d))).astype(np.float32)
    bank[1000:1400] = bank[1000]                               # 400 exact duplicates
    q = (u[None, :] + 0.02 * rng.standard_normal((nq, d))).astype(np.float32)       # every similarity of these queries is negative
    q[5] = 0.0                                                 # zero query: NaN scores
    q[6:40] = bank[1000] * np.float32(3.0)                     # queries whose best rows are the duplicates (ties -> larger row first)
    nn = make_bank(nnm, bank)
    idx, sims, cnt = nn.search_batch(q, k, mode=nnm.MODE_MFMA)
    assert nn.last_stats()[1] == nnm.MODE_MFMA
    oi, os_, oc = pyoracle.nns_search(bank, q, k)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
    assert np.all(os_[40:, 0] < 0.0) and list(oi[6]) == [1399, 1398, 1397, 1396, 1395, 1394]
    lim = np.minimum(np.arange(nq, dtype=np.int64) // 3, n)     # queries 0..17 see fewer than k rows, query 0..2 none
    idx, sims, cnt = nn.search_batch(q, k, row_limit=lim, mode=nnm.MODE_MFMA)
    oi, os_, oc = pyoracle.nns_search(bank, q, k, row_limit=lim)
    assert_topk_equal(idx, sims, cnt, oi, os_, oc, 1e-12)
