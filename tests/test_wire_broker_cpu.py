"""CPU: the packed descriptor wire path and the broker against golden vectors recorded from the
reference (oracle/gen_golden_wire.py -> tests/golden/wire_g10.npz) and the reference's own
property tests for the broker (tests/test_broker.py:43-120,213-265 re-expressed)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN
from cslam_amd.algebraic_connectivity_maximization import EdgeInterRobot
from cslam_amd.broker import Broker
from cslam_amd.wire import DescriptorChunk, PackedDescriptorBuffer, unknown_rows


@pytest.fixture(scope="module")
def g10():
    g = np.load(os.path.join(GOLDEN, "wire_g10.npz"))
    return json.loads(str(g["chunks"])), json.loads(str(g["broker"]))


def test_chunks_equal_reference_dict_to_list_chunks(g10):
    cases, _ = g10
    assert len(cases) == 12
    rng = np.random.default_rng(0)
    for c in cases:
        buf = PackedDescriptorBuffer(robot_id=3)
        order = rng.permutation(len(c["keys"]))                   # insertion order must not matter
        for i in order:
            buf.append(c["keys"][i], np.full(8, c["keys"][i], dtype=np.float64))
        got = buf.chunks(c["start"], c["size"])
        assert [list(map(int, ch.keyframe_ids)) for ch in got] == c["out"]
        for ch in got:
            assert ch.descriptors.dtype == np.float32 and np.array_equal(ch.descriptors[:, 0], ch.keyframe_ids)


def test_buffer_delete_and_overwrite_semantics():
    buf = PackedDescriptorBuffer(robot_id=0, capacity=2)
    for k in range(10):
        buf.append(k, np.arange(4) + k)
    buf.append(4, np.zeros(4))                                     # dict assignment overwrites
    assert len(buf) == 10 and np.all(buf.descriptors[4] == 0)
    assert buf.delete_below(-1) == 0 and len(buf) == 10            # below the first key: untouched
    assert buf.delete_below(6) == 6 and buf.first_key() == 6 and buf.last_key() == 9
    assert np.array_equal(buf.descriptors[0], np.arange(4) + 6)
    with pytest.raises(ValueError):
        buf.append(11, np.zeros(5))


def test_wire_precision_contract_and_round_trip():
    rng = np.random.default_rng(1)
    emb = rng.standard_normal((5, 64))                             # float64 embeddings
    buf = PackedDescriptorBuffer(robot_id=2)
    buf.extend(range(100, 105), emb)
    (ch,) = buf.chunks(0, 10)
    # reference: embedding.tolist() -> float32[] field -> np.asarray -> float64
    ref = np.asarray([np.float32(x) for x in emb[3].tolist()], dtype=np.float32).astype(np.float64)
    assert np.array_equal(ch.as_float64()[3], ref)
    back = DescriptorChunk.from_bytes(ch.to_bytes())
    assert back.robot_id == 2 and np.array_equal(back.keyframe_ids, ch.keyframe_ids)
    assert np.array_equal(back.descriptors, ch.descriptors) and ch.nbytes_payload() == 5 * 64 * 4
    msgs = back.messages()
    assert msgs[3].keyframe_id == 103 and np.array_equal(np.asarray(msgs[3].descriptor), ref)
    with pytest.raises(ValueError):
        DescriptorChunk.from_bytes(ch.to_bytes()[:-1])
    rows, last = unknown_rows(ch, 101)
    assert list(rows) == [2, 3, 4] and last == 104
    rows, last = unknown_rows(ch, 200)
    assert len(rows) == 0 and last == 200


def _is_cover(edges, involved_with_edges, vertices):
    for e in edges:
        if e.robot0_id in involved_with_edges and e.robot1_id in involved_with_edges:
            if (e.robot0_id, e.robot0_keyframe_id) not in vertices and \
                    (e.robot1_id, e.robot1_keyframe_id) not in vertices:
                return False
    return True


def test_broker_against_reference_golden(g10):
    _, cases = g10
    assert len(cases) == 19
    exact = 0
    for c in cases:
        edges = [EdgeInterRobot(*e) for e in c["edges"]]
        b = Broker(edges, c["involved"])
        assert b.is_multi_robot_graph == c["multi"]
        cover = b.brokerage(True)
        ref = [set(map(tuple, comp)) for comp in c["cover"]]
        assert len(cover) == len(ref)                               # same components (incl. isolated vertices)
        if not c["multi"]:
            assert cover == [] and b.brokerage(False) == []
            continue
        assert b.is_bipartite == c["bipartite"]
        flat = set().union(*cover)
        assert _is_cover(edges, set(b.robots_involved_with_edges), flat)
        comps = [set(b.vertices[i] for i in comp) for comp in b.components()]
        assert sorted(map(sorted, comps)) == sorted(sorted(x) for x in _ref_components(c))
        if c["bipartite"]:
            # minimum cover: same size per component; the set itself is the canonical Koenig cover for
            # one of the two ways of naming the sides (networkx names them by its own iteration order)
            other = b.vertex_cover(left_is_first=False)
            for mine, alt, comp in zip(cover, other, comps):
                r = [x for x in ref if x <= comp and len(x) == len(mine)]
                assert any(x == mine or x == alt for x in r), (mine, alt, r)
            assert sum(map(len, cover)) == sum(map(len, ref))
        else:
            # 2-approximation: never more than twice a maximal matching, like the reference's
            assert sum(map(len, cover)) <= 2 * _maximal_matching_size(edges, set(b.robots_involved_with_edges))
        exact += sorted(map(sorted, cover)) == sorted(map(sorted, ref))
        np.random.seed(c["dialog_seed"])
        dialog = b.brokerage(False)
        assert [sorted(list(v) for v in s) for s in dialog] == c["dialog"]   # same draws, same order
    assert exact >= 4


def _ref_components(c):
    """Components of the matching graph recomputed independently (union-find) for the comparison."""
    involved = set(c["involved"])
    with_edges = set()
    for r0, _, r1, _, _ in c["edges"]:
        with_edges |= {r for r in (r0, r1) if r in involved}
    parent = {}

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for r0, k0, r1, k1, _ in c["edges"]:
        for v in ((r0, k0), (r1, k1)):
            if v[0] in with_edges:
                parent.setdefault(v, v)
        if r0 in with_edges and r1 in with_edges:
            parent[find((r0, k0))] = find((r1, k1))
    groups = {}
    for v in parent:
        groups.setdefault(find(v), set()).add(v)
    return list(groups.values())


def _maximal_matching_size(edges, ok):
    used, m = set(), 0
    for e in edges:
        a, b = (e.robot0_id, e.robot0_keyframe_id), (e.robot1_id, e.robot1_keyframe_id)
        if a[0] in ok and b[0] in ok and a not in used and b not in used and a != b:
            used |= {a, b}
            m += 1
    return max(m, 1)


@pytest.mark.parametrize("nrob,ne,nkf,seed", [(2, 2000, 800, 1), (2, 300, 300, 2), (4, 1500, 400, 3)])
def test_broker_properties_at_scale(nrob, ne, nkf, seed):
    """tests/test_broker.py:43-120 re-expressed: valid cover, never more vertices than one per edge;
    bipartite: cover size equals the maximum matching size (Koenig)."""
    rng = np.random.default_rng(seed)
    edges = []
    for _ in range(ne):
        r0, r1 = rng.choice(nrob, size=2, replace=False)
        edges.append(EdgeInterRobot(int(r0), int(rng.integers(nkf)), int(r1), int(rng.integers(nkf)), 1.0))
    b = Broker(edges, list(range(nrob)))
    for use_cover in (True, False):
        comps = b.brokerage(use_cover)
        flat = set().union(*comps)
        assert sum(map(len, comps)) == len(flat)                   # components do not overlap
        assert _is_cover(edges, set(range(nrob)), flat)
        assert len(flat) <= len(set((tuple(e[:2]), tuple(e[2:4])) for e in edges))
    if nrob == 2:
        import networkx as nx                                       # independent check of minimality
        g = nx.Graph()
        for e in edges:
            g.add_edge((e.robot0_id, e.robot0_keyframe_id), (e.robot1_id, e.robot1_keyframe_id))
        top = {v for v in g if v[0] == 0}
        mm = nx.bipartite.maximum_matching(g, top_nodes=top)
        assert sum(map(len, b.brokerage(True))) == len(mm) // 2
