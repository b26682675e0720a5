"""Candidate selection in the reference's NORMAL operating regime (a few thousand poses, small budgets) with the solver 'auto' now
picks on a GPU host (chain_hip: cslam_mac_fw_subset / cslam_fiedler) against the reference's path (tracemin_lu: TraceMIN + SuperLU
on the host, what 'auto' used below 20 000 poses until round 2).  python tools/perf_acm_small.py"""
import random, sys, time
import numpy as np
sys.path.insert(0, ".")
from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot

R = 8
for P, C, K in ((250, 200, 5), (250, 2000, 100), (1000, 2000, 5), (1000, 2000, 100), (2000, 2000, 100), (2000, 5000, 200)):
    rnd = random.Random(P + C)
    fixed = [EdgeInterRobot(r, P - 1, r + 1, P - 1, 1.0) for r in range(R - 1)]
    cand = {}
    while len(cand) < C:
        a = rnd.randrange(R); b = rnd.choice([x for x in range(R) if x != a])
        e = EdgeInterRobot(a, rnd.randrange(P), b, rnd.randrange(P), round(0.1 + 0.9 * rnd.random(), 6))
        cand[(min(a, b), e.robot0_keyframe_id if a < b else e.robot1_keyframe_id, max(a, b),
              e.robot1_keyframe_id if a < b else e.robot0_keyframe_id)] = e
    cand = list(cand.values())
    res = {}
    for s in ("chain_hip", "tracemin_lu", "chain_hip"):          # chain_hip twice: the first call of a process pays library start-up
        params = {"frontend.enable_sparsification": True, "evaluation.enable_sparsification_comparison": False,
                  "frontend.mac_fiedler_solver": s}
        ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=R, extra_params=params)
        ac.set_graph(list(fixed), list(cand))
        t0 = time.perf_counter()
        sel = ac.select_candidates(K, {r: True for r in range(R)})
        res[s] = (time.perf_counter() - t0, sorted(tuple(e)[:4] for e in sel))
    same = res["chain_hip"][1] == res["tracemin_lu"][1]
    print(f"{R} robots x {P} poses = {R*P} poses, {C} candidates, budget {K}: chain_hip {res['chain_hip'][0]:.2f} s, "
          f"tracemin_lu (reference path, host) {res['tracemin_lu'][0]:.2f} s, same selection: {same}", flush=True)
