"""GPU parity of the lidar ScanContext path (csrc/scancontext.hip through the C ABI) against the
golden vectors recorded from the reference and against the CPU oracle (bit for bit)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, synth_scancontexts, synth_sc_revisits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g9():
    return np.load(os.path.join(GOLDEN, "sc_g9.npz"))


def new_matcher(**kw):
    from cslam_amd.lidar_pr.scancontext_matching import ScanContextMatching
    return ScanContextMatching(**kw)


def test_empty_matcher_answers_like_the_reference():
    m = new_matcher()
    assert m.search(np.zeros(1200), 1) == ([None], [None])
    assert m.search_best(np.zeros(1200)) == (None, None)
    assert m.scancontexts.shape == (1000, 20, 60) and m.ringkeys.shape == (1000, 20)


@pytest.mark.parametrize("name", ["n3", "n12", "n150", "n150c4"])
def test_golden_reference_cases(g9, name):
    bank = g9[name + "/bank_u16"].astype(np.float64) / 256.0
    q = g9[name + "/q_u16"].astype(np.float64) / 256.0
    n, ncand = len(bank), int(g9[name + "/ncand"])
    m = new_matcher(num_candidates=ncand)
    for i in range(n):
        m.add_item(bank[i].reshape(-1), 1000 + 7 * i)
    assert m.nb_items == n
    assert np.array_equal(m.ringkeys[:n], g9[name + "/ringkeys"])          # bit-identical to np.mean
    assert np.array_equal(m.scancontexts[:n], bank)
    for j in range(len(q)):
        items, sims = m.search(q[j].reshape(-1), 1)
        assert items == [int(g9[name + "/items"][j])]
        assert abs(sims[0] - g9[name + "/sims"][j]) <= 1e-12
        it, s = m.search_best(q[j].reshape(-1))
        assert it == items[0] and s == sims[0]
    d = m.search_diagnostics(q)
    ref_c = g9[name + "/cands"].copy()
    ref_c[ref_c >= n] = -1
    assert np.array_equal(d["cand"], ref_c)
    ok = ref_c >= 0
    assert np.abs(d["cdist"] - g9[name + "/dists"])[ok].max() <= 1e-12
    assert np.array_equal(d["cyaw"][ok], g9[name + "/yaws"][ok])


@pytest.mark.parametrize("n,nq,shape,ncand,seed", [(5000, 96, (20, 60), 10, 1), (2300, 40, (8, 16), 64, 2),
                                                   (700, 30, (33, 7), 3, 3), (40, 16, (20, 60), 64, 4)])
def test_bit_exact_vs_oracle(n, nq, shape, ncand, seed):
    from oracle import pyoracle
    rng = np.random.default_rng(seed)
    bank = synth_scancontexts(rng, n, *shape)
    q, place, shift = synth_sc_revisits(rng, bank, nq)
    q[-1] = 0.0
    q[-2] = synth_scancontexts(rng, 1, *shape)[0]
    q = q + (rng.random(q.shape) * 1e-3) * (q > 0)          # off the 1/256 grid: rounding order matters
    lim = rng.integers(0, n + 1, size=nq)
    lim[: nq // 2] = n
    lim[-3] = 0
    m = new_matcher(shape=list(shape), num_candidates=ncand)
    m.add_items(bank[: n // 3], range(n // 3))             # crosses the 1000-item growth boundary
    m.add_items(bank[n // 3:], range(n // 3, n))
    for row_limit in (None, lim):
        d = m.search_diagnostics(q, row_limit=row_limit)
        o = pyoracle.sc_search(bank, q, ncand, row_limit=row_limit)
        for key in ("cand", "cyaw", "best_idx", "best_yaw"):
            assert np.array_equal(d[key], o[key]), key
        assert np.array_equal(d["cdist"], o["cdist"])       # same fma chains: bit for bit
        assert np.array_equal(d["best_sim"], o["best_sim"])
    found = (d["best_idx"][: nq // 2] == place[: nq // 2]).mean()
    assert found > 0.9                                      # revisits are recovered


def test_revisit_yaw_recovered_and_batch_equals_sequential():
    rng = np.random.default_rng(9)
    bank = synth_scancontexts(rng, 300)
    q, place, shift = synth_sc_revisits(rng, bank, 24, noise=0.0)
    m = new_matcher()
    m.add_items(bank, [f"kf{i}" for i in range(300)])
    rows, sims, cnt = m.search_batch(q)
    assert np.array_equal(rows[:, 0], place) and np.all(cnt == 1)
    assert np.all(np.abs(sims[:, 0] - 1.0) < 1e-12)
    for j in range(24):
        items, s = m.search(q[j].reshape(-1), 1)
        assert items == [f"kf{place[j]}"] and s[0] == sims[j, 0]
        assert m.last_yaw_diff_deg == (shift[j] if shift[j] else 60) * 6.0
    # reference convention: no candidate under distance 1 -> first item, similarity 0.0
    items, s = m.search(np.zeros(1200), 1)
    assert items == ["kf0"] and s == [0.0]
    rows, sims, cnt = m.search_batch(np.zeros((2, 1200)), row_limit=np.array([0, 5]))
    assert list(cnt) == [0, 1] and rows[1, 0] == 0 and sims[1, 0] == 0.0 and rows[0, 0] == -1


def test_lidar_sparse_matching_batched_equals_sequential():
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    params = {"robot_id": 0, "max_nb_robots": 3, "frontend.similarity_threshold": 0.5,
              "frontend.sensor_type": "lidar", "frontend.nb_best_matches": 10,
              "frontend.intra_loop_min_inbetween_keyframes": 5, "frontend.enable_sparsification": True,
              "evaluation.enable_sparsification_comparison": False}
    rng = np.random.default_rng(21)
    places = synth_scancontexts(rng, 40)
    walk = rng.integers(0, 40, size=90)

    def obs(p):
        return synth_sc_revisits(np.random.default_rng(1000 + int(p) + int(rng.integers(1 << 30))), places[[p]], 1)[0][0]

    local = np.stack([obs(p) for p in walk[:50]]).reshape(50, -1)
    remote = np.stack([obs(p) for p in walk[50:]]).reshape(40, -1)

    class Msg:
        def __init__(self, r, k, d):
            self.robot_id, self.keyframe_id, self.descriptor = r, k, d.astype(np.float32).tolist()

    a = LoopClosureSparseMatching(params)
    b = LoopClosureSparseMatching(params)
    seq_intra, seq_inter = [], []
    for j in range(20):
        seq_inter.append(a.add_other_robot_global_descriptor(Msg(1, j, remote[j])))
    for j in range(50):
        kf, _ = a.match_local_loop_closures(local[j], j)
        seq_intra.append((j, kf))
        seq_inter.extend(a.add_local_global_descriptor(local[j], j))
    for j in range(20, 40):
        seq_inter.append(a.add_other_robot_global_descriptor(Msg(1, j, remote[j])))
    seq_inter = [e for e in seq_inter if e is not None]

    wire = remote.astype(np.float32).astype(np.float64)
    bat_inter = list(b.process_remote_descriptors(1, wire[:20], range(20)))
    bi, be = b.process_local_keyframes(local, range(50))
    bat_inter += be
    bat_inter += b.process_remote_descriptors(1, wire[20:], range(20, 40))
    assert bi == seq_intra
    assert [tuple(e) for e in bat_inter] == [tuple(e) for e in seq_inter]
    assert len(seq_inter) > 10 and sum(k is not None for _, k in seq_intra) > 5
    assert a.candidate_selector.candidate_edges.keys() == b.candidate_selector.candidate_edges.keys()


# ---- descriptor (ptcloud2sc) --------------------------------------------------------------
def test_ptcloud2sc_golden_reference_frames():
    from cslam_amd.lidar_pr.scancontext import ScanContext
    g = np.load(os.path.join(GOLDEN, "sc_cloud_g11.npz"))
    ex = ScanContext({}, None)
    for name in g["names"]:
        pts = g[name + "/pts"]
        d = ex.compute_embedding(pts)                           # float32 cloud, widened exactly
        assert d.shape == (1200,) and d.dtype == np.float64
        assert np.array_equal(d.reshape(20, 60), g[name + "/sc"]), name
    batch = ex.compute_embeddings([g[n + "/pts"].astype(np.float64) for n in g["names"]])
    for i, name in enumerate(g["names"]):
        assert np.array_equal(batch[i].reshape(20, 60), g[name + "/sc"])


@pytest.mark.parametrize("n,dense,seed", [(150000, True, 41), (1023, False, 42), (1025, False, 43), (0, False, 44)])
def test_ptcloud2sc_bit_exact_vs_oracle(n, dense, seed):
    from cslam_amd.lidar_pr.scancontext import ScanContext
    from helpers import synth_lidar_cloud
    from oracle import pyoracle
    pts = synth_lidar_cloud(np.random.default_rng(seed), n, dense).astype(np.float64) if n else np.zeros((0, 3))
    if n:
        pts += np.random.default_rng(seed).random(pts.shape) * 1e-7      # genuine float64 coordinates
    d = ScanContext({}, None).compute_embedding(pts).reshape(20, 60)
    assert np.array_equal(d, pyoracle.ptcloud2sc(pts))


def test_lidar_extract_then_match_round_trip():
    """A place seen twice, the second time with the sensor yawed by 90 degrees: the descriptors match with
    the yaw recovered (15 sectors of 6 degrees), through the two drop-in classes."""
    from cslam_amd.lidar_pr.scancontext import ScanContext
    from helpers import synth_lidar_cloud
    ex = ScanContext({}, None)
    m = new_matcher()
    clouds = [synth_lidar_cloud(np.random.default_rng(50 + i), 20000, False).astype(np.float64) for i in range(12)]
    for i, d in enumerate(ex.compute_embeddings(clouds)):
        m.add_item(d, i)
    c = clouds[7][~np.isnan(clouds[7]).any(axis=1)]
    rot = np.stack([-c[:, 1], c[:, 0], c[:, 2]], axis=1)               # +90 degrees about z
    items, sims = m.search(ex.compute_embedding(rot), 1)
    assert items == [7] and sims[0] > 0.9
    assert m.last_yaw_diff_deg in (45 * 6.0, 15 * 6.0)
