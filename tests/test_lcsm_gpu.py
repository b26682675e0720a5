"""GPU parity of the sparse-matching orchestrator (-m gpu): replays the 3-robot causal
sequence recorded from the REFERENCE LoopClosureSparseMatching (tests/golden/seq_g2.npz:
intra search -> add -> inter best-1, remote descriptors arriving as float64 lists) through
cslam_amd.loop_closure_sparse_matching, per-keyframe API and batched API, plus the
reference's own unit tests (tests/test_sparse_matching.py) re-expressed.
"""
from collections import namedtuple

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
GlobalDescriptor = namedtuple("GlobalDescriptor", ["keyframe_id", "robot_id", "descriptor"])


def make_params(robot_id, R, thr):
    return {"robot_id": robot_id, "max_nb_robots": R, "frontend.sensor_type": "stereo",
            "frontend.similarity_threshold": thr, "frontend.nb_best_matches": 10,
            "frontend.intra_loop_min_inbetween_keyframes": 20,
            "frontend.enable_sparsification": True, "evaluation.enable_sparsification_comparison": False}


def check_matches(got, ref):
    got = np.array(got, dtype=np.float64).reshape(-1, 6)
    assert got.shape == ref.shape
    assert np.array_equal(got[:, :5], ref[:, :5])                  # robots / keyframes identical
    assert np.max(np.abs(got[:, 5] - ref[:, 5])) < 1e-5 if len(ref) else True


@pytest.mark.parametrize("thr", [0.0, 0.1])
def test_sequence_replay_per_keyframe_api(thr):
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    g = np.load(GOLDEN + "/seq_g2.npz")
    tag = f"thr{thr}"
    desc = g[tag + "/desc"]
    R, T, D = desc.shape
    lcsms = [LoopClosureSparseMatching(make_params(r, R, thr)) for r in range(R)]
    intra, inter_local, inter_remote = [], [], []
    for t in range(T):
        for r in range(R):
            emb = desc[r, t]
            kf, kfs = lcsms[r].match_local_loop_closures(emb, t)
            intra.append((r, t, -1 if kf is None else kf))
            for m in lcsms[r].add_local_global_descriptor(emb, t):
                inter_local.append((r,) + tuple(m))
            msg = GlobalDescriptor(t, r, emb.tolist())
            for o in range(R):
                if o != r:
                    m = lcsms[o].add_other_robot_global_descriptor(msg)
                    if m is not None:
                        inter_remote.append((o,) + tuple(m))
    assert np.array_equal(np.array(intra, dtype=np.int64), g[tag + "/intra"])
    check_matches(inter_local, g[tag + "/inter_local"])
    check_matches(inter_remote, g[tag + "/inter_remote"])
    for r in range(R):
        keys = sorted(lcsms[r].candidate_selector.candidate_edges.keys())
        assert np.array_equal(np.array(keys, dtype=np.int64).reshape(-1, 4), g[tag + f"/cand_keys_r{r}"])
        w = np.array([lcsms[r].candidate_selector.candidate_edges[k].weight for k in keys])
        assert np.max(np.abs(w - g[tag + f"/cand_w_r{r}"])) < 1e-5


def test_sequence_replay_batched_api():
    """Robot 0 ingests its 120 keyframes in 4 batches; remote descriptors arrive in between.
    Must give the same intra matches as the per-keyframe golden, and inter matches equal to a
    sequential replay with the same arrival order."""
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    g = np.load(GOLDEN + "/seq_g2.npz")
    desc = g["thr0.1/desc"]
    R, T, D = desc.shape
    a = LoopClosureSparseMatching(make_params(0, R, 0.1))
    b = LoopClosureSparseMatching(make_params(0, R, 0.1))
    seq_intra, seq_inter, bat_intra, bat_inter = [], [], [], []
    for s in range(0, T, 30):
        ids = list(range(s, s + 30))
        for t in ids:
            kf, _ = a.match_local_loop_closures(desc[0, t], t)
            seq_intra.append((t, kf))
            seq_inter += a.add_local_global_descriptor(desc[0, t], t)
        if (s // 30) % 2:                       # every other batch through the two halves a pipelining host uses, with
            import torch                         # unrelated GPU work (the next chunk's extraction) enqueued in between
            h = b.process_local_keyframes_begin(desc[0, s:s + 30], ids)
            filler = torch.randn(2048, 2048, device="cuda")
            filler = filler @ filler
            i2, m2 = h.finish()
            assert h.finish() == (i2, m2)
        else:
            i2, m2 = b.process_local_keyframes(desc[0, s:s + 30], ids)
        bat_intra += i2
        bat_inter += m2
        for o in (1, 2):
            for t in ids:
                m = a.add_other_robot_global_descriptor(GlobalDescriptor(t, o, desc[o, t].tolist()))
                if m is not None:
                    seq_inter.append(m)
            bat_inter += b.process_remote_descriptors(o, desc[o, s:s + 30].astype(np.float64), ids)
    assert seq_intra == bat_intra
    ref_intra = g["thr0.1/intra"]
    ref0 = [(int(t), None if k < 0 else int(k)) for r, t, k in ref_intra if r == 0]
    assert ref0 == seq_intra
    assert [tuple(m)[:4] for m in seq_inter] == [tuple(m)[:4] for m in bat_inter]
    assert np.max(np.abs(np.array([m.weight for m in seq_inter]) - np.array([m.weight for m in bat_inter]))) < 1e-12
    assert sorted(a.candidate_selector.candidate_edges) == sorted(b.candidate_selector.candidate_edges)


def test_reference_unit_tests_reexpressed():
    """tests/test_sparse_matching.py of the reference: stored descriptors, cosine==Euclid
    ordering, best-match identity, budgeted selection sizes."""
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    from cslam_amd.nns_matching import NearestNeighborsMatching
    rng = np.random.default_rng(0)
    p = make_params(0, 2, 0.0)
    lcsm = LoopClosureSparseMatching(p)
    d = rng.random(10); d /= np.linalg.norm(d)
    lcsm.add_local_global_descriptor(d, 1)
    assert np.allclose(lcsm.local_nnsm.data[0], d, atol=1e-7)
    lcsm.add_other_robot_global_descriptor(GlobalDescriptor(0, 1, d.tolist()))
    assert np.allclose(lcsm.other_robots_nnsm[1].data[0], d, atol=1e-7)

    nnsm = NearestNeighborsMatching()
    for i in range(100):
        v = rng.random(100); v /= np.linalg.norm(v)
        nnsm.add_item(v, i)
    for _ in range(25):
        q = rng.random(100); q /= np.linalg.norm(q)
        ds = np.linalg.norm(q[None, :] - nnsm.data[:nnsm.n], axis=1)
        order = np.argsort(ds)[:100]
        ns, sims = nnsm.search(q, 100)
        assert np.all(sims[:-1] >= sims[1:])
        for j in range(100):
            if order[j] != ns[j]:
                assert abs(sims[order[j]] - sims[ns[j]]) < 1e-6 or abs(ds[order[j]] - ds[ns[j]]) < 1e-6
        assert nnsm.search_best(q)[0] == order[0]

    lcsm = LoopClosureSparseMatching(make_params(0, 2, 0.0))
    d0 = rng.random(10); d0 /= np.linalg.norm(d0)
    lcsm.add_local_global_descriptor(d0, 2)
    d1 = 1 - d0; d1 /= np.linalg.norm(d1)
    lcsm.add_other_robot_global_descriptor(GlobalDescriptor(3, 1, d1.tolist()))
    d2 = d0                     # the reference test mutates the array it added: the bank must hold a copy
    d2[0] = 0.0; d2[1] = 0.0
    d2 = d2 / np.linalg.norm(d2)
    lcsm.add_other_robot_global_descriptor(GlobalDescriptor(4, 1, d2.tolist()))
    rid = lcsm.candidate_selector.candidate_edges[(0, 2, 1, 4)].robot1_id
    assert np.allclose(lcsm.other_robots_nnsm[rid].data[0], d1, atol=1e-7)

    for R, budget in ((3, 10), (4, 10)):
        lcsm = LoopClosureSparseMatching(make_params(0, R, 0.0))
        for i in range(100):
            v = rng.random(10); v /= np.linalg.norm(v)
            lcsm.add_local_global_descriptor(v, i)
        for r in range(1, R):
            for i in range(100):
                v = rng.random(10); v /= np.linalg.norm(v)
                lcsm.add_other_robot_global_descriptor(GlobalDescriptor(i, r, v.tolist()))
        sel = lcsm.select_candidates(budget, {r: True for r in range(R)})
        assert len(sel) == budget


def test_packed_wire_path_equals_per_message_callbacks():
    """Robot 1 publishes through PackedDescriptorBuffer -> bytes -> DescriptorChunk; robot 0 ingests the
    chunks with process_remote_chunk.  Same matches as feeding robot 0 one GlobalDescriptor message per
    keyframe (gdlcd.py:407-422), including the re-sent rows that get_unknown_range drops."""
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    from cslam_amd.wire import DescriptorChunk, PackedDescriptorBuffer
    g = np.load(GOLDEN + "/seq_g2.npz")
    desc = g["thr0.1/desc"]
    R, T, D = desc.shape
    a = LoopClosureSparseMatching(make_params(0, R, 0.1))
    b = LoopClosureSparseMatching(make_params(0, R, 0.1))
    for x in (a, b):
        x.process_local_keyframes(desc[0], list(range(T)), intra=False)
    buf = PackedDescriptorBuffer(robot_id=1)
    seq, bat, last = [], [], -1
    sent = 0
    for upto in (25, 60, T):
        for t in range(sent, upto):
            buf.append(t, desc[1, t])
        resend_from, sent = max(0, sent - 15), upto
        for ch in buf.chunks(resend_from, 10):                 # overlaps what was sent before
            wire = DescriptorChunk.from_bytes(ch.to_bytes())
            assert len(wire) <= 10
            m, new_last = b.process_remote_chunk(wire, last)
            bat += m
            for msg in wire.messages():                         # reference: one callback row at a time
                if msg.keyframe_id > last:
                    r = a.add_other_robot_global_descriptor(msg)
                    if r is not None:
                        seq.append(r)
            last = new_last
    assert last == T - 1 and a.other_robots_nnsm[1].n == T and b.other_robots_nnsm[1].n == T
    assert [tuple(m) for m in seq] == [tuple(m) for m in bat] and len(seq) > 0
    assert np.array_equal(a.other_robots_nnsm[1].data, b.other_robots_nnsm[1].data)


def test_drained_queue_of_remote_messages_equals_one_message_at_a_time():
    """`process_remote_chunks` (several robots' messages in one call: one search of the local bank) against
    `process_remote_chunk` message by message: same matches, same last-received ids, same banks, same candidate
    edges -- with re-sent rows and messages that arrive before the local bank has a row."""
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    from cslam_amd.wire import DescriptorChunk, PackedDescriptorBuffer
    g = np.load(GOLDEN + "/seq_g2.npz")
    desc = g["thr0.1/desc"]
    R, T, D = desc.shape
    assert R >= 3 and T > 50
    a = LoopClosureSparseMatching(make_params(0, R, 0.1))
    b = LoopClosureSparseMatching(make_params(0, R, 0.1))
    bufs = {o: PackedDescriptorBuffer(robot_id=o) for o in range(1, R)}
    last_a = {o: -1 for o in bufs}
    last_b = dict(last_a)
    seq, bat = [], []
    lsent = rsent = 0
    for lupto, rupto in ((0, 5), (20, 25), (45, 50), (T, T)):   # first round: messages before the local bank has a row
        if lupto > lsent:
            for x in (a, b):
                x.process_local_keyframes(desc[0, lsent:lupto], list(range(lsent, lupto)), intra=False)
        msgs = []
        for o, buf in bufs.items():
            for t in range(rsent, rupto):
                buf.append(t, desc[o, t])
            for ch in buf.chunks(max(0, rsent - 7 * o), 16):     # robot o re-sends its last 7 o rows
                msgs.append(DescriptorChunk.from_bytes(ch.to_bytes()))
        lsent, rsent = lupto, rupto
        for wire in msgs:
            m, last_a[wire.robot_id] = a.process_remote_chunk(wire, last_a[wire.robot_id])
            seq += [tuple(e) for e in m]
        # ONE call for the whole drained queue, several messages per robot included: a robot's second message is filtered
        # with the id its first one left, whatever (stale) id the caller supplies for it
        res = b.process_remote_chunks([(w, last_b[w.robot_id]) for w in msgs])
        for w, (m, new_last) in zip(msgs, res):
            last_b[w.robot_id] = max(last_b[w.robot_id], new_last)
            bat += [tuple(e) for e in m]
    assert last_a == last_b and all(v == T - 1 for v in last_a.values())
    for o in bufs:
        assert a.other_robots_nnsm[o].n == b.other_robots_nnsm[o].n == T
        assert np.array_equal(a.other_robots_nnsm[o].data, b.other_robots_nnsm[o].data)
    # same matches (the calls reorder messages of different robots only).  The similarity of a pair may differ in its last
    # bits: a message with <= 8 new rows takes the exact scan, the same rows inside a larger batch the MFMA search whose
    # float64 rescoring sums in another order
    def same(x, y):
        x, y = sorted(x), sorted(y)
        return len(x) == len(y) and all(p[:4] == q[:4] and abs(p[4] - q[4]) <= 1e-12 for p, q in zip(x, y))
    assert len(seq) > 0 and same(seq, bat)
    ea = [tuple(e) for e in a.candidate_selector.candidate_edges.values()]
    eb = [tuple(e) for e in b.candidate_selector.candidate_edges.values()]
    assert len(ea) > 0 and same(ea, eb)
