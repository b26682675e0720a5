"""oracle/gen_golden_mac.py -- TEST INFRASTRUCTURE (build container only).

G7: seeded multi-robot graphs through the REFERENCE AlgebraicConnectivityMaximization
    (greedy_initialization=True) with MAC.evaluate_fiedler_pair wrapped to record
    lambda_2 and ||grad|| of every Frank-Wolfe iteration; stores the selected edges.
G8: a scripted add_match / select / candidate_edges_to_fixed / remove sequence; stores the
    candidate keys, offsets and rekeyed edges after every step.
Called from gen_golden.py (stubs installed there).
"""
import os
import random

import numpy as np


def _graph(R, P, C, seed):
    """Fixed chain of inter-robot links (so MAC, not the greedy fallback, runs) + C candidates."""
    rnd = random.Random(seed)
    fixed = [(r, P - 1, r + 1, P - 1, 1.0) for r in range(R - 1)]
    cand = {}
    while len(cand) < C:
        if R == 1:
            a, b = 0, 0
        else:
            a = rnd.randrange(R)
            b = rnd.choice([x for x in range(R) if x != a])
        e = (a, rnd.randrange(P), b, rnd.randrange(P), round(0.1 + 0.9 * rnd.random(), 6))
        key = (e[0], e[1], e[2], e[3]) if e[0] < e[2] else (e[2], e[3], e[0], e[1])
        if R == 1 and abs(e[1] - e[3]) < 2:
            continue
        cand[key] = e
    return fixed, list(cand.values())


def gen_mac(out_dir):
    from cslam.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot
    from cslam.mac.mac import MAC

    g = {}
    cases = [(1, 100, 50, 10, 0), (3, 100, 100, 10, 1), (5, 100, 200, 100, 2), (8, 400, 600, 60, 3)]
    for (R, P, C, K, seed) in cases:
        fixed, cand = _graph(R, P, C, seed)
        trace = []
        orig = MAC.evaluate_fiedler_pair

        def wrapped(self, w, method='tracemin_lu', tol=1e-8, _orig=orig, _trace=trace):
            f, v = _orig(self, w, method, tol)
            _trace.append((float(f), float(np.linalg.norm(self.grad_from_fiedler(v)))))
            return f, v

        MAC.evaluate_fiedler_pair = wrapped
        try:
            ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=R)
            ac.set_graph([EdgeInterRobot(*e) for e in fixed], [EdgeInterRobot(*e) for e in cand])
            sel = ac.select_candidates(K, {r: True for r in range(R)}, greedy_initialization=True)
        finally:
            MAC.evaluate_fiedler_pair = orig
        tag = f"mac_R{R}_P{P}_C{C}_K{K}"
        g[tag + "/fixed"] = np.array(fixed, dtype=np.float64).reshape(-1, 5)
        g[tag + "/cand"] = np.array(cand, dtype=np.float64).reshape(-1, 5)
        g[tag + "/selected"] = np.array([tuple(e) for e in sel], dtype=np.float64).reshape(-1, 5)
        g[tag + "/trace"] = np.array(trace, dtype=np.float64).reshape(-1, 2)
        g[tag + "/remaining"] = np.array(sorted(ac.candidate_edges.keys()), dtype=np.int64).reshape(-1, 4)
        lam = f"lambda2 {trace[0][0]:.6g} -> {trace[-1][0]:.6g}" if trace else "(greedy path: no fixed inter-robot link)"
        print(f"  {tag}: selected {len(sel)}, FW iterations {len(trace)}, {lam}")

    # ---- G8: bookkeeping script (3 robots; this robot = 1)
    R = 3
    ac = AlgebraicConnectivityMaximization(robot_id=1, max_nb_robots=R)
    rnd = random.Random(11)
    log = []

    def snap(step):
        keys = sorted(ac.candidate_edges.keys())
        log.append((step, keys, [ac.candidate_edges[k].weight for k in keys], dict(ac.nb_poses),
                    dict(ac.initial_fixed_edge_exists), len(ac.fixed_edges),
                    sorted(ac.already_considered_matches)))

    matches = []
    for t in range(60):
        a = rnd.randrange(R)
        b = rnd.choice([x for x in range(R) if x != a])
        m = (a, rnd.randrange(30), b, rnd.randrange(30), round(rnd.random(), 6))
        matches.append(m)
        ac.add_match(EdgeInterRobot(*m))
    for m in matches[:10]:                                 # same edges again, other weights/direction
        ac.add_match(EdgeInterRobot(m[2], m[3], m[0], m[1], m[4] + 0.5))
        ac.add_match(EdgeInterRobot(m[0], m[1], m[2], m[3], m[4] - 0.5))
    snap(0)
    inr = {0: True, 1: True, 2: True}
    sel0 = ac.select_candidates(5, inr)                    # no fixed links yet -> biased greedy
    snap(1)
    ac.candidate_edges_to_fixed(list(sel0[:3]))
    ac.remove_candidate_edges(list(sel0[3:]), failed=True)
    snap(2)
    sel1 = ac.select_candidates(4, inr)                    # MAC (if both others now linked) or greedy
    snap(3)
    sel2 = ac.select_candidates(4, {0: True, 1: True, 2: False})
    snap(4)
    ac.compute_offsets(ac.check_graph_disconnections({0: True, 1: True, 2: False}))
    g["acm/matches"] = np.array(matches, dtype=np.float64)
    for i, s in enumerate((sel0, sel1, sel2)):
        g[f"acm/sel{i}"] = np.array([tuple(e) for e in s], dtype=np.float64).reshape(-1, 5)
    for (step, keys, w, nbp, ife, nfixed, acm) in log:
        g[f"acm/s{step}_keys"] = np.array(keys, dtype=np.int64).reshape(-1, 4)
        g[f"acm/s{step}_w"] = np.array(w, dtype=np.float64)
        g[f"acm/s{step}_nb_poses"] = np.array([nbp[r] for r in range(R)], dtype=np.int64)
        g[f"acm/s{step}_ife"] = np.array([ife[r] for r in range(R)], dtype=np.bool_)
        g[f"acm/s{step}_nfixed"] = np.int64(nfixed)
        g[f"acm/s{step}_considered"] = np.array(acm, dtype=np.int64).reshape(-1, 4)
    g["acm/final_offsets"] = np.array([ac.offsets[r] for r in range(R)], dtype=np.int64)
    print("  acm script: selections", [len(s) for s in (sel0, sel1, sel2)])
    np.savez_compressed(os.path.join(out_dir, "mac_g7.npz"), **g)
