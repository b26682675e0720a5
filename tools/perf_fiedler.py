"""One Fiedler pair through the C ABI's cslam_fiedler at pose-graph scale (GPU box):
python tools/perf_fiedler.py [poses_per_robot] [loop_edges] [repeats]      (CSLAM_MAC_TIMING=1 prints the phases)"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_mac_gpu import _pose_graph
from cslam_amd.mac.chain_solver_gpu import fiedler_tracemin_hip

P = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 3
fiedler_tracemin_hip(_pose_graph(2, 500, 10, 0))            # library start-up
L = _pose_graph(8, P, m, 1)
for _ in range(rep):
    st = {}
    t0 = time.perf_counter(); l2, v2 = fiedler_tracemin_hip(L, stats=st); t1 = time.perf_counter()
    print(f"n={L.shape[0]} loop_edges={m}: cslam_fiedler lambda2={l2:.6e} {st['iters']} iterations, {t1-t0:.3f}s", flush=True)
