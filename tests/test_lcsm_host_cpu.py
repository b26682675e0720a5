"""Host bookkeeping of the batched matcher (no GPU): the array forms equal the reference's per-keyframe /
per-match forms (cslam/loop_closure_sparse_matching.py:74-92, cslam/algebraic_connectivity_maximization.py:559-572)."""
import numpy as np

from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot
from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching


def _params(r, R=4, thr=0.5, gap=5):
    return {"robot_id": r, "max_nb_robots": R, "frontend.sensor_type": "stereo", "frontend.similarity_threshold": thr,
            "frontend.nb_best_matches": 6, "frontend.intra_loop_min_inbetween_keyframes": gap,
            "frontend.enable_sparsification": True, "evaluation.enable_sparsification_comparison": False}


def test_add_matches_arrays_equals_sequential_add_match():
    rng = np.random.default_rng(3)
    for me in (0, 2, 3):                         # robot0 < robot1, mixed, robot0 > robot1: the un-normalised-key quirk
        a = AlgebraicConnectivityMaximization(me, 4)
        b = AlgebraicConnectivityMaximization(me, 4)
        for rnd in range(6):
            n = 400
            kf0 = rng.integers(0, 40, size=n)
            r1 = rng.choice([r for r in range(4) if r != me], size=n)
            kf1 = rng.integers(0, 30, size=n)                       # plenty of repeated keys inside and across batches
            w = np.round(rng.random(n), 2)                          # and equal weights
            seq = [EdgeInterRobot(me, int(x), int(y), int(z), np.float64(v)) for x, y, z, v in zip(kf0, r1, kf1, w)]
            for e in seq:
                a.add_match(e)
            out = b.add_matches_arrays(me, kf0, r1, kf1, w)
            assert [tuple(e) for e in out] == [tuple(e) for e in seq]
            assert list(a.candidate_edges.keys()) == list(b.candidate_edges.keys())          # same insertion order
            assert [tuple(e) for e in a.candidate_edges.values()] == [tuple(e) for e in b.candidate_edges.values()]
            assert a.nb_poses == b.nb_poses
            # move some to "already considered" (selected / failed) and keep going
            gone = list(a.candidate_edges.values())[::7]
            a.remove_candidate_edges(gone)
            b.remove_candidate_edges(gone)
        assert a.already_considered_matches == b.already_considered_matches


def test_add_matches_arrays_empty_and_same_robot():
    a = AlgebraicConnectivityMaximization(1, 3)
    assert a.add_matches_arrays(1, [], [], [], []) == []
    b = AlgebraicConnectivityMaximization(1, 3)
    out = a.add_matches_arrays(1, [3, 4], [1, 2], [5, 6], [0.7, 0.8])      # one intra-robot edge: general path
    for e in [EdgeInterRobot(1, 3, 1, 5, 0.7), EdgeInterRobot(1, 4, 2, 6, 0.8)]:
        b.add_match(e)
    assert a.candidate_edges == b.candidate_edges and a.nb_poses == b.nb_poses and len(out) == 2


def test_intra_decision_arrays_equal_the_reference_loop():
    rng = np.random.default_rng(5)
    lc = LoopClosureSparseMatching(_params(0))
    m, k, n = 300, 6, 500
    items = rng.permutation(n).astype(np.int64)                     # row -> keyframe id
    rows = rng.integers(0, n, size=(m, k))
    sims = np.sort(rng.random((m, k)), axis=1)[:, ::-1].copy()
    sims[rng.random((m, k)) < 0.05] = np.nan
    cnt = rng.integers(0, k + 1, size=m).astype(np.int32)
    ids = rng.integers(0, n, size=m).astype(np.int64)
    for j in range(0, m, 3):                                        # the keyframe itself heads its own list
        if cnt[j] > 0:
            rows[j, 0] = np.nonzero(items == ids[j])[0][0]
    for j in range(1, m, 5):                                        # ... or sits further down (must NOT be stripped)
        if cnt[j] > 2:
            rows[j, 2] = np.nonzero(items == ids[j])[0][0]
    got = lc._intra_from_topk(rows, sims, cnt, ids, items)
    gap, thr = 5, 0.5
    for j in range(m):
        kfs = [int(items[r]) for r in rows[j, :cnt[j]]]
        s = list(sims[j, :cnt[j]])
        if kfs and kfs[0] == ids[j]:
            kfs, s = kfs[1:], s[1:]
        want = lc._first_valid(kfs, s, int(ids[j]), gap, thr) if kfs else None
        assert got[j] == (int(ids[j]), want), (j, got[j], want)


def test_drained_queue_filters_a_robots_second_message_with_the_id_its_first_one_left():
    """process_remote_chunks with two messages of one robot in a call == two callbacks in a row
    (neighbors_manager.py:147-169 keeps the last id per robot between callbacks): the stale id supplied with the
    second message must not let re-sent rows through.  Host-only form (a bank without `search_device`)."""
    from cslam_amd.wire import DescriptorChunk

    class _Bank(object):                         # no search_device: the per-message path
        n = 0

    lcsm = LoopClosureSparseMatching.__new__(LoopClosureSparseMatching)
    lcsm.params = _params(0)
    lcsm.local_nnsm = _Bank()
    seen = []

    def fake_chunk(chunk, last):
        from cslam_amd.wire import unknown_rows
        rows, new_last = unknown_rows(chunk, last)
        seen.append((int(chunk.robot_id), int(last), [int(chunk.keyframe_ids[r]) for r in rows]))
        return [], new_last
    lcsm.process_remote_chunk = fake_chunk
    d = np.zeros((4, 8), dtype=np.float32)
    m1 = DescriptorChunk(1, np.array([0, 1, 2, 3], dtype=np.int32), d)
    m2 = DescriptorChunk(1, np.array([2, 3, 4, 5], dtype=np.int32), d)       # overlaps m1
    m3 = DescriptorChunk(2, np.array([0, 1, 2, 3], dtype=np.int32), d)
    res = lcsm.process_remote_chunks([(m1, -1), (m3, 1), (m2, -1)])
    assert [r[1] for r in res] == [3, 3, 5]
    assert seen == [(1, -1, [0, 1, 2, 3]), (2, 1, [2, 3]), (1, 3, [4, 5])]
