"""Shared test helpers: synthetic-descriptor recipe and golden access."""
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def unit_rows(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return x


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def nns_case_names(g):
    return sorted({k.split("/")[0] for k in g.files})


def nns_case_inputs(g, name):
    """Inputs of a golden NNS case: stored, or regenerated from the seeds used by
    oracle/gen_golden.py (sha256-checked so RNG drift cannot go unnoticed)."""
    if name + "/bank" in g.files:
        bank, q = g[name + "/bank"], g[name + "/queries"]
    elif name.startswith("c1_s"):
        seed = int(name[4])
        bank = unit_rows(np.random.default_rng(1234 + seed), 1000, 4096)
        q = unit_rows(np.random.default_rng(4321 + seed), 64, 4096)
        if name.endswith("f64"):
            q = q.astype(np.float64)
    elif name.startswith("r_n"):
        parts = name.split("_")
        n, d = int(parts[1][1:]), int(parts[2][1:])
        seed = {(257, 512): 2, (257, 64): 3, (1, 64): 4, (33, 128): 5, (300, 4096): 7}[(n, d)]
        bank = unit_rows(np.random.default_rng(100 + seed), n, d)
        if name.endswith("f64"):
            q = np.random.default_rng(300 + seed).standard_normal((16, d))
        else:
            q = unit_rows(np.random.default_rng(200 + seed), 16, d)
    else:
        raise KeyError(name)
    assert sha(bank.astype(np.float32)) == str(g[name + "/bank_sha"]), "bank RNG drift: " + name
    assert sha(q) == str(g[name + "/q_sha"]), "query RNG drift: " + name
    return bank, q


def assert_topk_equal(idx, sims, cnt, ridx, rsims, rcnt, score_tol):
    assert np.array_equal(cnt, rcnt)
    assert np.array_equal(idx, ridx), f"top-k indices differ at {np.argwhere(idx != ridx)[:5]}"
    m = ridx >= 0
    if m.any():
        a, b = sims[m], rsims[m]
        both_nan = np.isnan(a) & np.isnan(b)
        assert np.all(both_nan | (np.abs(a - b) <= score_tol)), float(np.nanmax(np.abs(a - b)))


# ---- lidar ScanContext synthetic recipe (heights quantised to 1/256 so fixtures store u16) ----
def synth_scancontexts(rng, n, rings=20, sectors=60):
    """n random scan contexts [n, rings, sectors] float64: heights in [0, 6), ~30 % empty bins,
    ~10 % empty sectors (columns), quantised to multiples of 1/256."""
    h = rng.random((n, rings, sectors)) * 6.0
    h[rng.random((n, rings, sectors)) < 0.3] = 0.0
    h *= (rng.random((n, 1, sectors)) >= 0.1)
    return np.floor(h * 256.0) / 256.0


def synth_sc_revisits(rng, bank, m, noise=0.15):
    """m revisit queries: a bank place rotated by a random sector shift, jittered on its non-empty
    bins.  Returns (queries [m, rings, sectors], place [m], shift [m])."""
    n, _, sectors = bank.shape
    place = rng.integers(0, n, size=m)
    shift = rng.integers(0, sectors, size=m)
    q = np.empty((m,) + bank.shape[1:])
    for j in range(m):
        s = np.roll(bank[place[j]], int(shift[j]), axis=1)
        s = np.where(s > 0, np.maximum(s + noise * rng.standard_normal(s.shape), 0.0), 0.0)
        q[j] = np.floor(s * 256.0) / 256.0
    return q, place, shift


def synth_lidar_cloud(rng, n, dense_core=True):
    """n lidar-like points [n, 3] float32 (exactly representable in float64): ranges skewed to the near
    field so that some polar bins collect far more than 500 returns (the cap in the reference's
    ptcloud2sc), heights around the -2 m ground plane, a few NaN returns and exact zeros."""
    r = rng.gamma(2.0, 6.0 if dense_core else 15.0, size=n)
    a = rng.random(n) * 2 * np.pi
    if dense_core:
        a[: n // 4] = rng.normal(0.3, 0.02, size=n // 4)       # a wall: many returns in few sectors
        r[: n // 4] = rng.normal(5.0, 0.3, size=n // 4)
    z = -2.0 + np.abs(rng.normal(0.0, 1.2, size=n)) * (rng.random(n) < 0.7) - 0.3 * (rng.random(n) < 0.1)
    pts = np.stack([r * np.cos(a), r * np.sin(a), z], axis=1).astype(np.float32)
    k = max(n // 200, 3)
    pts[rng.integers(0, n, size=k), rng.integers(0, 3, size=k)] = np.nan
    pts[rng.integers(0, n, size=3), 0] = 0.0
    pts[rng.integers(0, n, size=3), 1] = 0.0
    return pts[rng.permutation(n)]
