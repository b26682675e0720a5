#!/usr/bin/env python
"""oracle/gen_golden_wire.py -- TEST INFRASTRUCTURE.  Generates tests/golden/wire_g10.npz.

Runs ONLY in the build container: imports the real reference from /root/reference and records
  * cslam/utils/misc.py:21-32 `dict_to_list_chunks` on seeded key sets (the chunking of the
    descriptor publication buffer, gdlcd.py:198-227), and
  * cslam/broker.py `Broker.brokerage` (vertex cover and simple dialog) on seeded edge lists.
The fixture holds inputs and the reference's outputs only (json strings inside the npz).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import _install_stubs  # noqa: E402


def main():
    _install_stubs()
    from sortedcontainers import SortedDict
    from cslam.utils.misc import dict_to_list_chunks
    from cslam.broker import Broker
    from cslam.algebraic_connectivity_maximization import EdgeInterRobot

    rng = np.random.default_rng(10)
    chunk_cases = []
    for t in range(12):
        n = int(rng.integers(1, 40))
        keys = sorted(set(int(k) for k in rng.integers(0, 60, size=n)))
        if t % 3 == 0:
            keys = list(range(keys[0], keys[0] + len(keys)))          # the usual contiguous ids
        d = SortedDict((k, k) for k in keys)
        start = int(rng.integers(-3, 65))
        size = int(rng.integers(1, 12))
        chunk_cases.append(dict(keys=keys, start=start, size=size, out=dict_to_list_chunks(d, start, size)))

    broker_cases = []
    for t in range(16):
        nrob = 2 if t < 8 else int(rng.integers(3, 6))
        involved = list(range(nrob)) if t % 4 else list(range(nrob + 1))
        ne = int(rng.integers(1, 40))
        nkf = int(rng.integers(3, 25))
        edges = []
        for _ in range(ne):
            r0, r1 = rng.choice(nrob + (1 if t % 5 == 0 else 0), size=2, replace=False)
            edges.append((int(r0), int(rng.integers(nkf)), int(r1), int(rng.integers(nkf)), float(rng.random())))
        el = [EdgeInterRobot(*e) for e in edges]
        b = Broker(el, involved)
        cover = b.brokerage(True)
        np.random.seed(100 + t)
        dialog = b.brokerage(False)
        broker_cases.append(dict(edges=edges, involved=involved,
                                 multi=bool(b.is_multi_robot_graph),
                                 bipartite=bool(getattr(b, "is_bipartite", False)),
                                 cover=[sorted([list(v) for v in c]) for c in cover],
                                 dialog_seed=100 + t,
                                 dialog=[sorted([list(v) for v in c]) for c in dialog]))
        print(t, "robots", nrob, "edges", ne, "components", len(cover), "cover size", sum(len(c) for c in cover),
              "dialog size", sum(len(c) for c in dialog))
    # degenerate inputs
    for edges, involved in (([], [0, 1]), ([(0, 1, 1, 2, 0.5)], [0]), ([(0, 1, 1, 2, 0.5)], [2, 3])):
        b = Broker([EdgeInterRobot(*e) for e in edges], involved)
        broker_cases.append(dict(edges=edges, involved=involved, multi=bool(b.is_multi_robot_graph), bipartite=False,
                                 cover=[sorted([list(v) for v in c]) for c in b.brokerage(True)],
                                 dialog_seed=0, dialog=[sorted([list(v) for v in c]) for c in b.brokerage(False)]))
    path = os.path.join(HERE, "..", "tests", "golden", "wire_g10.npz")
    np.savez_compressed(path, chunks=json.dumps(chunk_cases), broker=json.dumps(broker_cases))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
