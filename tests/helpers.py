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
