"""Randomised shapes through the batched search (-m gpu): for 64 seeded draws of (bank rows, dimension, batch, k, dtype, row limits,
row / query scales) the MFMA-mode result (fp16-pair candidate stage + float64 re-scoring + certificate) must equal the exact float64
scan kernel on every query and the CPU oracle (oracle/nns_oracle.c, the restatement of cslam/nns_matching.py:42-61) on a sample --
indices identical, scores within 1e-12.  Dimensions that are not multiples of 32, batches that are not multiples of a tile, banks
that end inside a tile, k up to 16, causal limits, float64 queries, clustered rows (near-ties)."""
import numpy as np
import pytest

from helpers import assert_topk_equal
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def _draw(seed):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([300, 1000, 2571, 5000, 9001, 20000]))
    d = int(rng.choice([17, 64, 100, 128, 257, 512, 1000, 4096]))
    nq = int(rng.choice([9, 70, 129, 255, 257, 600, 1500]))
    if n * d > 3e7:
        n = int(3e7 // d)
    k = int(rng.choice([1, 3, 5, 10, 16]))
    f64 = bool(rng.integers(0, 2))
    kind = int(rng.integers(0, 4))
    bank = rng.standard_normal((n, d)).astype(np.float32)
    if kind == 1:                                    # rows of very different norms, values spread inside a row
        bank *= np.exp2(rng.integers(-20, 20, size=(n, 1))).astype(np.float32)
        bank *= np.exp2(rng.integers(-6, 1, size=(n, d))).astype(np.float32)
    elif kind == 2:                                  # clusters: many near-ties around every query's best rows
        centres = rng.standard_normal((max(n // 50, 1), d)).astype(np.float32)
        bank = centres[rng.integers(0, len(centres), size=n)] + 1e-3 * rng.standard_normal((n, d)).astype(np.float32)
    elif kind == 3:                                  # unit-norm descriptors, the extractors' output
        bank /= np.linalg.norm(bank, axis=1, keepdims=True)
    q = rng.standard_normal((nq, d))
    if kind == 2:
        q = bank[rng.integers(0, n, size=nq)].astype(np.float64) + 1e-3 * rng.standard_normal((nq, d))
    q *= np.exp2(rng.integers(-10, 10, size=(nq, 1)))
    q = q.astype(np.float64 if f64 else np.float32)
    lim = None
    if rng.integers(0, 3) == 0:
        lim = rng.integers(0, n + 1, size=nq).astype(np.int64)
    return bank, q, k, lim


@pytest.mark.parametrize("seed", range(64))
def test_random_shapes_mfma_mode_equals_scan_and_oracle(seed):
    from cslam_amd import nns_matching as nnm
    bank, q, k, lim = _draw(seed)
    nn = nnm.NearestNeighborsMatching()
    nn.add_items(bank, range(len(bank)))
    idx, sims, cnt = nn.search_batch(q, k, row_limit=lim, mode=nnm.MODE_MFMA)
    assert nn.last_stats()[1] == nnm.MODE_MFMA
    idx2, sims2, cnt2 = nn.search_batch(q, k, row_limit=lim, mode=nnm.MODE_SCAN)
    assert_topk_equal(idx, sims, cnt, idx2, sims2, cnt2, 1e-12)
    sel = np.random.default_rng(seed).choice(len(q), size=min(24, len(q)), replace=False)
    oi, os_, oc = pyoracle.nns_search(bank, q[sel], k, row_limit=None if lim is None else lim[sel])
    assert_topk_equal(idx[sel], sims[sel], cnt[sel], oi, os_, oc, 1e-12)
