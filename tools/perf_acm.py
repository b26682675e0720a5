"""Whole candidate selection at pose-graph scale (GPU box):
python tools/perf_acm.py [poses_per_robot] [candidates] [budget] [solvers comma-separated]"""
import random, sys, time
import numpy as np
sys.path.insert(0, ".")
from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot

P = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
solvers = (sys.argv[4] if len(sys.argv) > 4 else "chain_gpu").split(",")
R = 8
rnd = random.Random(0)
fixed = [EdgeInterRobot(r, P - 1, r + 1, P - 1, 1.0) for r in range(R - 1)]
cand = {}
while len(cand) < C:
    a = rnd.randrange(R); b = rnd.choice([x for x in range(R) if x != a])
    e = EdgeInterRobot(a, rnd.randrange(P), b, rnd.randrange(P), round(0.1 + 0.9 * rnd.random(), 6))
    cand[(min(a, b), e.robot0_keyframe_id if a < b else e.robot1_keyframe_id, max(a, b),
          e.robot1_keyframe_id if a < b else e.robot0_keyframe_id)] = e
cand = list(cand.values())
sels = {}
from cslam_amd.mac.mac import MAC
_orig = MAC.evaluate_fiedler_pair
_times = []
def _timed(self, w, method='tracemin_lu', tol=1e-8):
    t0 = time.perf_counter(); L = self.combined_laplacian(w); t1 = time.perf_counter()
    out = self.find_fiedler_pair(L, method, tol); t2 = time.perf_counter()
    _times.append((t1 - t0, t2 - t1, int((w > 1e-10).sum())))
    return out
MAC.evaluate_fiedler_pair = _timed
for s in solvers:
    params = {"frontend.enable_sparsification": True, "evaluation.enable_sparsification_comparison": False,
              "frontend.mac_fiedler_solver": s}
    ac = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=R, extra_params=params)
    ac.set_graph(list(fixed), list(cand))
    t0 = time.perf_counter()
    sel = ac.select_candidates(K, {r: True for r in range(R)})
    dt = time.perf_counter() - t0
    sels[s] = sorted(tuple(e)[:4] for e in sel)
    if _times:
        print('   per FW iteration (laplacian s, fiedler s, active edges):', [(round(a, 2), round(b, 2), c) for a, b, c in _times[:3]], '...', [(round(a, 2), round(b, 2), c) for a, b, c in _times[-2:]], 'sum fiedler %.1fs' % sum(b for _, b, _ in _times))
        _times.clear()
    print(f"{s}: n={R*P} candidates={C} K={K}: select_candidates {dt:.1f}s, selected {len(sel)}", flush=True)
if len(sels) > 1:
    base = sels[solvers[0]]
    for s in solvers[1:]:
        common = len(set(base) & set(sels[s]))
        print(f"   selection overlap {solvers[0]} vs {s}: {common}/{len(base)}")
