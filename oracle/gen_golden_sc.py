#!/usr/bin/env python
"""oracle/gen_golden_sc.py -- TEST INFRASTRUCTURE.  Generates tests/golden/sc_g9.npz.

Runs ONLY in the build container: imports the real reference
(cslam/lidar_pr/scancontext_matching.py, scancontext_utils.py) from /root/reference and records,
for seeded synthetic scan contexts, what ScanContextMatching.search / search_best return plus the
intermediate quantities (KD-tree ring-key candidates, per-candidate distance_sc distance and yaw).
The fixture holds inputs (u16-quantised heights) and the reference's outputs only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, "/root/reference")
from helpers import synth_scancontexts, synth_sc_revisits  # noqa: E402


def main():
    from scipy import spatial
    import cslam.lidar_pr.scancontext_utils as sc_utils
    from cslam.lidar_pr.scancontext_matching import ScanContextMatching

    out = {}
    names = []
    for name, seed, n, m, ncand in (("n3", 11, 3, 3, 10), ("n12", 12, 12, 4, 10),
                                    ("n150", 13, 150, 8, 10), ("n150c4", 14, 150, 4, 4)):
        rng = np.random.default_rng(seed)
        bank = synth_scancontexts(rng, n)
        q, place, shift = synth_sc_revisits(rng, bank, m)
        q[-1] = synth_scancontexts(rng, 1)[0]            # one query that revisits nothing
        if name == "n12":
            q[0] = 0.0                                    # all-empty query -> "no match" branch
        matcher = ScanContextMatching(num_candidates=ncand)
        for i in range(n):
            matcher.add_item(bank[i].reshape(-1), 1000 + 7 * i)
        items, sims, cands, dists, yaws = [], [], [], [], []
        for j in range(m):
            it, s = matcher.search(q[j].reshape(-1), 1)
            it2, s2 = matcher.search_best(q[j].reshape(-1))
            assert it2 == it[0] and s2 == s[0]
            items.append(it[0]); sims.append(s[0])
            tree = spatial.KDTree(np.array(matcher.ringkeys[:n]))
            _, ci = tree.query(sc_utils.sc2rk(q[j]), k=ncand)
            ci = np.atleast_1d(ci)
            cands.append(ci)
            dd, yy = [], []
            for c in ci:
                d, y = sc_utils.distance_sc(matcher.scancontexts[c], q[j])
                dd.append(d); yy.append(y)
            dists.append(dd); yaws.append(yy)
        out[name + "/bank_u16"] = np.round(bank * 256.0).astype(np.uint16)
        out[name + "/q_u16"] = np.round(q * 256.0).astype(np.uint16)
        out[name + "/ringkeys"] = np.array(matcher.ringkeys[:n])
        out[name + "/items"] = np.array(items, dtype=np.int64)
        out[name + "/sims"] = np.array(sims, dtype=np.float64)
        out[name + "/cands"] = np.array(cands, dtype=np.int64)
        out[name + "/dists"] = np.array(dists, dtype=np.float64)
        out[name + "/yaws"] = np.array(yaws, dtype=np.int64)
        out[name + "/place"] = place
        out[name + "/shift"] = shift
        out[name + "/ncand"] = np.int64(ncand)
        names.append(name)
        print(name, "items", items, "sims", np.round(sims, 4), "place", 1000 + 7 * place, "shift", shift)
    # empty matcher behaviour (scancontext_matching.py:52-53, 96-97)
    e = ScanContextMatching()
    r = e.search(np.zeros(1200), 1)
    assert r == ([None], [None]) and e.search_best(np.zeros(1200)) == (None, None)
    out["names"] = np.array(names)
    path = os.path.join(HERE, "..", "tests", "golden", "sc_g9.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main_cloud():
    """ptcloud2sc (scancontext_utils.py:46-75) on float64 point clouds -> tests/golden/sc_cloud_g11.npz."""
    import cslam.lidar_pr.scancontext_utils as sc_utils
    from helpers import synth_lidar_cloud
    out = {}
    for name, seed, n, dense in (("wall40k", 31, 40000, True), ("sparse6k", 32, 6000, False), ("tiny", 33, 40, False)):
        pts32 = synth_lidar_cloud(np.random.default_rng(seed), n, dense)
        pts = pts32.astype(np.float64)
        sc = sc_utils.ptcloud2sc(pts, [20, 60], 80)
        out[name + "/pts"] = pts32
        out[name + "/sc"] = sc
        cnt = np.zeros((20, 60), int)
        for p in pts:
            if not np.isnan(p).any():
                r, c = sc_utils.pt2rs(p, 4.0, 6.0, 20, 60)
                cnt[r, c] += 1
        print(name, "points", n, "bins over the 500 cap:", int((cnt > 500).sum()), "max", cnt.max(),
              "nonzero bins", int((sc != 0).sum()))
    out["names"] = np.array(["wall40k", "sparse6k", "tiny"])
    path = os.path.join(HERE, "..", "tests", "golden", "sc_cloud_g11.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if "cloud" in sys.argv[1:]:
        main_cloud()
    else:
        main()
